// Minimal YAML subset reader for the reference's config files (DCReg/config/*.yaml, loaded by yaml-cpp in
// DCReg/src/icp_test_runner.cpp:20-153): nested block maps by indentation, scalars, quoted strings/keys,
// inline flow lists [a, b], '#' comments.  No anchors, no multi-line scalars, no block lists.
#pragma once
#include <cstdlib>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace yamlite {

struct Node {
    bool is_map = false;
    std::string scalar;
    std::vector<std::string> list;                       // flow list, if any
    bool is_list = false;
    std::map<std::string, std::shared_ptr<Node>> kids;   // std::map: lexicographic order, as runAllTests iterates (:306)

    bool has(const std::string &k) const { return kids.count(k) > 0; }
    const Node &operator[](const std::string &k) const {
        auto it = kids.find(k);
        if (it == kids.end()) throw std::runtime_error("yaml: missing key '" + k + "'");
        return *it->second;
    }
    double as_double() const {
        char *end = nullptr;
        const double v = std::strtod(scalar.c_str(), &end);
        if (end == scalar.c_str()) throw std::runtime_error("yaml: not a number: '" + scalar + "'");
        return v;
    }
    int as_int() const { return (int)as_double(); }
    bool as_bool() const { return scalar == "true" || scalar == "True" || scalar == "yes" || scalar == "1"; }
    const std::string &as_string() const { return scalar; }
};

inline std::string trim(const std::string &s) {
    size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}
inline std::string unquote(const std::string &s) {
    if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\''))) return s.substr(1, s.size() - 2);
    return s;
}
inline std::string strip_comment(const std::string &line) {
    bool in_s = false, in_d = false;
    for (size_t i = 0; i < line.size(); ++i) {
        const char c = line[i];
        if (c == '"' && !in_s) in_d = !in_d;
        else if (c == '\'' && !in_d) in_s = !in_s;
        else if (c == '#' && !in_s && !in_d && (i == 0 || line[i - 1] == ' ' || line[i - 1] == '\t')) return line.substr(0, i);
    }
    return line;
}

inline Node parse(std::istream &in) {
    Node root; root.is_map = true;
    struct Level { int indent; Node *node; };
    std::vector<Level> stack{{-1, &root}};
    std::string raw;
    while (std::getline(in, raw)) {
        const std::string line = strip_comment(raw);
        if (trim(line).empty()) continue;
        const int indent = (int)line.find_first_not_of(" \t");
        const std::string body = trim(line);
        // split key: value at the first ':' outside quotes
        size_t colon = std::string::npos;
        bool in_s = false, in_d = false;
        for (size_t i = 0; i < body.size(); ++i) {
            const char c = body[i];
            if (c == '"' && !in_s) in_d = !in_d;
            else if (c == '\'' && !in_d) in_s = !in_s;
            else if (c == ':' && !in_s && !in_d && (i + 1 == body.size() || body[i + 1] == ' ')) { colon = i; break; }
        }
        if (colon == std::string::npos) throw std::runtime_error("yaml: cannot parse line: " + raw);
        const std::string key = unquote(trim(body.substr(0, colon)));
        const std::string val = trim(body.substr(colon + 1));
        while (stack.size() > 1 && indent <= stack.back().indent) stack.pop_back();
        Node *parent = stack.back().node;
        auto node = std::make_shared<Node>();
        if (val.empty()) {
            node->is_map = true;
            parent->kids[key] = node;
            stack.push_back({indent, node.get()});
        } else if (val.front() == '[') {
            node->is_list = true;
            const size_t close = val.rfind(']');
            std::string inner = val.substr(1, close == std::string::npos ? std::string::npos : close - 1);
            std::string item; std::stringstream ss(inner);
            while (std::getline(ss, item, ',')) { item = unquote(trim(item)); if (!item.empty()) node->list.push_back(item); }
            parent->kids[key] = node;
        } else {
            node->scalar = unquote(val);
            parent->kids[key] = node;
        }
    }
    return root;
}

inline Node load_file(const std::string &path) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error("yaml: cannot open " + path);
    return parse(f);
}

}  // namespace yamlite
