// xml_mini.hpp — small XML DOM reader for the COLLADA ingest.
// Stands in for the reference's `parseval::xml` dependency (git dep, not vendored):
// an element is either Data (text) or Elements (children); an empty element counts as
// Elements([]), which is what the bundled scenes need (`<library_images/>`).
#pragma once
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace mi355rt {

struct XmlElement {
    std::string name;
    std::vector<std::pair<std::string, std::string>> attribs;
    std::vector<std::unique_ptr<XmlElement>> children;
    std::string data;
    bool has_data = false;          // true: DataOrElements::Data, false: ::Elements

    const std::string* attrib(const std::string& key) const
    {
        for (auto& a : attribs) if (a.first == key) return &a.second;
        return nullptr;
    }
    const XmlElement* child_by_name(const std::string& n) const
    {
        for (auto& c : children) if (c->name == n) return c.get();
        return nullptr;
    }
    const XmlElement* child_by_attrib(const std::string& key, const std::string& value) const
    {
        for (auto& c : children) {
            const std::string* v = c->attrib(key);
            if (v && *v == value) return c.get();
        }
        return nullptr;
    }
};

struct XmlDoc {
    bool has_definition = false;    // <?xml ... ?>
    std::unique_ptr<XmlElement> root;
    std::string remaining;          // non-whitespace text after the root's closing tag
};

// Returns false with a message on malformed input.
bool xml_parse(const std::string& text, XmlDoc& doc, std::string& err);

}  // namespace mi355rt
