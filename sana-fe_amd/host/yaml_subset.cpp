// yaml_subset.cpp -- see yaml_subset.hpp.
#include "yaml_subset.hpp"

#include <fstream>
#include <sstream>
#include <stdexcept>

namespace sanafe_amd
{
namespace
{
[[noreturn]] void fail(const std::string &what, int line)
{
    throw std::invalid_argument("YAML: " + what + " (line " + std::to_string(line) + ")");
}

struct Line
{
    int indent;
    std::string text; // without indentation, comment and trailing blanks
    int number;
};

std::string rtrim(std::string s)
{
    while (!s.empty() && (s.back() == ' ' || s.back() == '\t' || s.back() == '\r')) s.pop_back();
    return s;
}
std::string trim(const std::string &s)
{
    size_t b = 0;
    while (b < s.size() && (s[b] == ' ' || s[b] == '\t')) b++;
    return rtrim(s.substr(b));
}

// Removes a trailing comment: '#' at line start or preceded by a blank, outside quotes.
std::string strip_comment(const std::string &s)
{
    bool sq = false, dq = false;
    for (size_t i = 0; i < s.size(); i++)
    {
        const char c = s[i];
        if (c == '\'' && !dq) sq = !sq;
        else if (c == '"' && !sq) dq = !dq;
        else if (c == '#' && !sq && !dq && (i == 0 || s[i - 1] == ' ' || s[i - 1] == '\t')) return s.substr(0, i);
    }
    return s;
}

int bracket_balance(const std::string &s)
{
    int depth = 0;
    bool sq = false, dq = false;
    for (char c : s)
    {
        if (c == '\'' && !dq) sq = !sq;
        else if (c == '"' && !sq) dq = !dq;
        else if (!sq && !dq)
        {
            if (c == '[' || c == '{') depth++;
            else if (c == ']' || c == '}') depth--;
        }
    }
    return depth;
}

// Splits the text into logical lines; a flow collection that spans physical lines is joined.
std::vector<Line> logical_lines(const std::string &text)
{
    std::vector<Line> out;
    std::istringstream in(text);
    std::string raw;
    int n = 0;
    while (std::getline(in, raw))
    {
        n++;
        std::string s = rtrim(strip_comment(raw));
        size_t ind = 0;
        while (ind < s.size() && s[ind] == ' ') ind++;
        if (ind < s.size() && s[ind] == '\t') fail("tabs are not allowed for indentation", n);
        if (ind == s.size()) continue; // blank
        if (s.compare(ind, 3, "---") == 0 && ind == 0) continue; // document marker
        Line l{static_cast<int>(ind), s.substr(ind), n};
        int depth = bracket_balance(l.text);
        while (depth > 0)
        {
            if (!std::getline(in, raw)) fail("unterminated flow collection", l.number);
            n++;
            const std::string more = trim(strip_comment(raw));
            if (more.empty()) continue;
            l.text += " " + more;
            depth += bracket_balance(more);
        }
        if (depth < 0) fail("unbalanced brackets", l.number);
        out.push_back(std::move(l));
    }
    return out;
}

std::string unquote(const std::string &s)
{
    if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\''))) return s.substr(1, s.size() - 2);
    return s;
}

// ---------------- flow parser ----------------
struct Flow
{
    const std::string &s;
    size_t p{0};
    int line;
    void ws()
    {
        while (p < s.size() && (s[p] == ' ' || s[p] == '\t')) p++;
    }
    // plain/quoted scalar up to one of the stop characters at depth 0
    std::string token(const char *stops)
    {
        ws();
        std::string t;
        if (p < s.size() && (s[p] == '"' || s[p] == '\''))
        {
            const char q = s[p++];
            while (p < s.size() && s[p] != q) t += s[p++];
            if (p >= s.size()) fail("unterminated quoted string", line);
            p++;
            return t;
        }
        while (p < s.size())
        {
            const char c = s[p];
            bool stop = false;
            for (const char *k = stops; *k; k++)
                if (c == *k) stop = true;
            // ':' only separates when followed by a blank or the end (so "a.b:c" style plain scalars survive)
            if (c == ':' && !(p + 1 >= s.size() || s[p + 1] == ' ' || s[p + 1] == ',' || s[p + 1] == ']' || s[p + 1] == '}')) stop = false;
            if (stop) break;
            t += c;
            p++;
        }
        return trim(t);
    }
    YamlNode value()
    {
        ws();
        YamlNode n;
        n.line = line;
        if (p < s.size() && s[p] == '[')
        {
            p++;
            n.kind = YamlNode::Seq;
            ws();
            if (p < s.size() && s[p] == ']')
            {
                p++;
                return n;
            }
            for (;;)
            {
                YamlNode item = entry();
                n.seq.push_back(std::move(item));
                ws();
                if (p < s.size() && s[p] == ',')
                {
                    p++;
                    ws();
                    if (p < s.size() && s[p] == ']') // trailing comma
                    {
                        p++;
                        return n;
                    }
                    continue;
                }
                if (p < s.size() && s[p] == ']')
                {
                    p++;
                    return n;
                }
                fail("expected ',' or ']' in flow sequence", line);
            }
        }
        if (p < s.size() && s[p] == '{')
        {
            p++;
            n.kind = YamlNode::Map;
            ws();
            if (p < s.size() && s[p] == '}')
            {
                p++;
                return n;
            }
            for (;;)
            {
                std::string key = token(":,}");
                ws();
                YamlNode v;
                v.line = line;
                if (p < s.size() && s[p] == ':')
                {
                    p++;
                    v = value();
                }
                n.map.emplace_back(std::move(key), std::move(v));
                ws();
                if (p < s.size() && s[p] == ',')
                {
                    p++;
                    ws();
                    if (p < s.size() && s[p] == '}')
                    {
                        p++;
                        return n;
                    }
                    continue;
                }
                if (p < s.size() && s[p] == '}')
                {
                    p++;
                    return n;
                }
                fail("expected ',' or '}' in flow mapping", line);
            }
        }
        n.kind = YamlNode::Scalar;
        n.scalar = token(",]}:");
        if (n.scalar.empty() || n.scalar == "~" || n.scalar == "null") n.kind = n.scalar.empty() ? YamlNode::Null : YamlNode::Scalar;
        return n;
    }
    // an entry of a flow sequence: a value, or a single `key: value` pair (an implicit mapping)
    YamlNode entry()
    {
        ws();
        if (p < s.size() && (s[p] == '[' || s[p] == '{')) return value();
        const size_t save = p;
        std::string key = token(",]}:");
        ws();
        if (p < s.size() && s[p] == ':' && (p + 1 >= s.size() || s[p + 1] == ' ' || s[p + 1] == '[' || s[p + 1] == '{'))
        {
            p++;
            YamlNode m;
            m.kind = YamlNode::Map;
            m.line = line;
            m.map.emplace_back(std::move(key), value());
            return m;
        }
        p = save;
        return value();
    }
};

YamlNode parse_inline(const std::string &text, int line)
{
    const std::string t = trim(text);
    YamlNode n;
    n.line = line;
    if (t.empty()) return n;
    if (t[0] == '[' || t[0] == '{')
    {
        Flow f{t, 0, line};
        n = f.value();
        f.ws();
        if (f.p != t.size()) fail("trailing characters after flow collection", line);
        return n;
    }
    n.kind = YamlNode::Scalar;
    n.scalar = unquote(t);
    return n;
}

// position of the ": " (or trailing ':') that ends a block-mapping key, or npos
size_t key_colon(const std::string &s)
{
    bool sq = false, dq = false;
    int depth = 0;
    for (size_t i = 0; i < s.size(); i++)
    {
        const char c = s[i];
        if (c == '\'' && !dq) sq = !sq;
        else if (c == '"' && !sq) dq = !dq;
        else if (sq || dq) continue;
        else if (c == '[' || c == '{') depth++;
        else if (c == ']' || c == '}') depth--;
        else if (c == ':' && depth == 0 && (i + 1 == s.size() || s[i + 1] == ' ')) return i;
    }
    return std::string::npos;
}

struct Block
{
    std::vector<Line> lines;
    size_t i{0};

    YamlNode node(int indent)
    {
        if (i >= lines.size() || lines[i].indent < indent) return YamlNode{};
        const int ind = lines[i].indent;
        if (lines[i].text[0] == '-' && (lines[i].text.size() == 1 || lines[i].text[1] == ' ')) return sequence(ind);
        if (key_colon(lines[i].text) != std::string::npos && lines[i].text[0] != '[' && lines[i].text[0] != '{') return mapping(ind);
        YamlNode n = parse_inline(lines[i].text, lines[i].number);
        i++;
        return n;
    }
    YamlNode sequence(int ind)
    {
        YamlNode n;
        n.kind = YamlNode::Seq;
        n.line = lines[i].number;
        while (i < lines.size() && lines[i].indent == ind && lines[i].text[0] == '-' &&
                (lines[i].text.size() == 1 || lines[i].text[1] == ' '))
        {
            // rewrite "- rest" as a line holding "rest" at the column where it starts
            const std::string &t = lines[i].text;
            size_t k = 1;
            while (k < t.size() && t[k] == ' ') k++;
            if (k >= t.size())
            {
                i++;
                n.seq.push_back(node(ind + 1));
                continue;
            }
            lines[i].indent = ind + static_cast<int>(k);
            lines[i].text = t.substr(k);
            n.seq.push_back(node(lines[i].indent));
        }
        return n;
    }
    YamlNode mapping(int ind)
    {
        YamlNode n;
        n.kind = YamlNode::Map;
        n.line = lines[i].number;
        while (i < lines.size() && lines[i].indent == ind)
        {
            const Line &l = lines[i];
            if (l.text[0] == '-' && (l.text.size() == 1 || l.text[1] == ' ')) break; // a sequence at the same indent ends the mapping
            const size_t c = key_colon(l.text);
            if (c == std::string::npos) fail("expected 'key: value'", l.number);
            const std::string key = unquote(trim(l.text.substr(0, c)));
            const std::string rest = trim(l.text.substr(c + 1));
            const int line_no = l.number;
            i++;
            if (!rest.empty())
            {
                n.map.emplace_back(key, parse_inline(rest, line_no));
            }
            else if (i < lines.size() && (lines[i].indent > ind ||
                             (lines[i].indent == ind && lines[i].text[0] == '-' && (lines[i].text.size() == 1 || lines[i].text[1] == ' '))))
            {
                n.map.emplace_back(key, node(lines[i].indent)); // nested block (a sequence may sit at the key's indent)
            }
            else
            {
                n.map.emplace_back(key, YamlNode{});
            }
        }
        if (i < lines.size() && lines[i].indent > ind) fail("bad indentation", lines[i].number);
        return n;
    }
};

void json_string(std::ostringstream &o, const std::string &s)
{
    o << '"';
    for (char c : s)
    {
        if (c == '"' || c == '\\') o << '\\' << c;
        else if (c == '\n') o << "\\n";
        else o << c;
    }
    o << '"';
}
void to_json(std::ostringstream &o, const YamlNode &n)
{
    switch (n.kind)
    {
    case YamlNode::Null: o << "null"; break;
    case YamlNode::Scalar: json_string(o, n.scalar); break;
    case YamlNode::Seq:
        o << '[';
        for (size_t i = 0; i < n.seq.size(); i++)
        {
            if (i) o << ',';
            to_json(o, n.seq[i]);
        }
        o << ']';
        break;
    case YamlNode::Map:
        o << '{';
        for (size_t i = 0; i < n.map.size(); i++)
        {
            if (i) o << ',';
            json_string(o, n.map[i].first);
            o << ':';
            to_json(o, n.map[i].second);
        }
        o << '}';
        break;
    }
}
} // namespace

YamlNode yaml_parse(const std::string &text)
{
    Block b;
    b.lines = logical_lines(text);
    if (b.lines.empty()) return YamlNode{};
    YamlNode root = b.node(b.lines[0].indent);
    if (b.i != b.lines.size()) fail("unexpected content", b.lines[b.i].number);
    return root;
}

YamlNode yaml_parse_file(const std::string &path)
{
    std::ifstream f(path);
    if (!f) throw std::invalid_argument("cannot open " + path);
    std::ostringstream ss;
    ss << f.rdbuf();
    return yaml_parse(ss.str());
}

std::string yaml_to_json(const YamlNode &n)
{
    std::ostringstream o;
    to_json(o, n);
    return o.str();
}
} // namespace sanafe_amd
