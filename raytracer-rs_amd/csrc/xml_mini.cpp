// xml_mini.cpp — see xml_mini.hpp.
#include "xml_mini.hpp"
#include <cctype>

namespace mi355rt {
namespace {

struct Cursor {
    const std::string& s;
    size_t i = 0;
    explicit Cursor(const std::string& t) : s(t) {}
    bool eof() const { return i >= s.size(); }
    char peek() const { return s[i]; }
    bool starts(const char* lit) const { return s.compare(i, std::char_traits<char>::length(lit), lit) == 0; }
    void skip_ws() { while (!eof() && std::isspace((unsigned char)s[i])) ++i; }
};

bool is_name_char(char c)
{
    return std::isalnum((unsigned char)c) || c == '_' || c == '-' || c == ':' || c == '.';
}

void skip_misc(Cursor& c)     // whitespace and comments
{
    for (;;) {
        c.skip_ws();
        if (c.starts("<!--")) {
            size_t e = c.s.find("-->", c.i + 4);
            c.i = (e == std::string::npos) ? c.s.size() : e + 3;
        } else {
            return;
        }
    }
}

bool parse_element(Cursor& c, std::unique_ptr<XmlElement>& out, std::string& err, int depth)
{
    if (depth > 256) { err = "xml nesting too deep"; return false; }
    if (c.eof() || c.peek() != '<') { err = "expected '<' at offset " + std::to_string(c.i); return false; }
    ++c.i;
    auto el = std::make_unique<XmlElement>();
    size_t b = c.i;
    while (!c.eof() && is_name_char(c.peek())) ++c.i;
    if (c.i == b) { err = "expected element name at offset " + std::to_string(b); return false; }
    el->name = c.s.substr(b, c.i - b);
    // attributes
    for (;;) {
        c.skip_ws();
        if (c.eof()) { err = "unterminated element <" + el->name + ">"; return false; }
        if (c.starts("/>")) { c.i += 2; out = std::move(el); return true; }
        if (c.peek() == '>') { ++c.i; break; }
        size_t kb = c.i;
        while (!c.eof() && is_name_char(c.peek())) ++c.i;
        if (c.i == kb) { err = "bad attribute in <" + el->name + ">"; return false; }
        std::string key = c.s.substr(kb, c.i - kb);
        c.skip_ws();
        if (c.eof() || c.peek() != '=') { err = "attribute without value in <" + el->name + ">"; return false; }
        ++c.i;
        c.skip_ws();
        if (c.eof() || (c.peek() != '"' && c.peek() != '\'')) { err = "unquoted attribute in <" + el->name + ">"; return false; }
        char q = c.peek();
        ++c.i;
        size_t vb = c.i;
        while (!c.eof() && c.peek() != q) ++c.i;
        if (c.eof()) { err = "unterminated attribute in <" + el->name + ">"; return false; }
        el->attribs.emplace_back(key, c.s.substr(vb, c.i - vb));
        ++c.i;
    }
    // content
    std::string text;
    for (;;) {
        if (c.eof()) { err = "missing closing element </" + el->name + ">"; return false; }
        if (c.starts("<!--")) { skip_misc(c); continue; }
        if (c.starts("</")) {
            c.i += 2;
            size_t nb = c.i;
            while (!c.eof() && is_name_char(c.peek())) ++c.i;
            std::string close = c.s.substr(nb, c.i - nb);
            c.skip_ws();
            if (c.eof() || c.peek() != '>' || close != el->name) {
                err = "mismatched closing element </" + close + "> for <" + el->name + ">";
                return false;
            }
            ++c.i;
            break;
        }
        if (c.peek() == '<') {
            std::unique_ptr<XmlElement> child;
            if (!parse_element(c, child, err, depth + 1)) return false;
            el->children.push_back(std::move(child));
            continue;
        }
        text.push_back(c.peek());
        ++c.i;
    }
    if (el->children.empty()) {
        bool all_ws = true;
        for (char ch : text) if (!std::isspace((unsigned char)ch)) { all_ws = false; break; }
        if (!all_ws) { el->has_data = true; el->data = text; }
    }
    out = std::move(el);
    return true;
}

}  // namespace

bool xml_parse(const std::string& text, XmlDoc& doc, std::string& err)
{
    Cursor c(text);
    c.skip_ws();
    if (c.starts("<?xml")) {
        size_t e = text.find("?>", c.i);
        if (e == std::string::npos) { err = "unterminated xml definition"; return false; }
        doc.has_definition = true;
        c.i = e + 2;
    }
    skip_misc(c);
    if (!parse_element(c, doc.root, err, 0)) return false;
    skip_misc(c);
    doc.remaining = text.substr(c.i);
    return true;
}

}  // namespace mi355rt
