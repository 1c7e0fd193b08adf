// rtc_loader.cpp -- the callers either side of the hot path, in C++ like the reference's are compiled code:
//   * YAML scene files      (parse_scene/src/lib.rs:735-784 and everything it calls)
//   * Wavefront OBJ meshes  (obj_file/src/lib.rs:40-242)
//   * P3 PPM output         (ray_tracer/src/canvas.rs:44-89)
// They are clients of the boundary: every object is pushed through the same rtc_scene_* entry points a Rust
// host would call.  Quirks reproduced on purpose: Q11 (OBJ polygon fans reuse normals 0,1,2),
// Q14 (a `material:` on a YAML group is parsed, then dropped), Q16 (`extend` overrides only present keys;
// transform lists left-multiply in list order).
#include <cctype>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <limits>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

#include "rtc_scene.h"

using namespace rtc;

namespace {

struct ParseFail {
    std::string msg;
};
[[noreturn]] void fail(const std::string &m) { throw ParseFail{m}; }

// ------------------------------------------------------------------------------------------------
// A YAML subset sufficient for scene files (what serde_yaml::from_str::<Sequence> is asked to read there):
// block sequences and mappings nested by indentation, flow sequences / mappings, comments, plain and quoted
// scalars typed like serde_yaml 0.9 types them (see scalar()).
// ------------------------------------------------------------------------------------------------
struct Node {
    enum Kind { NUL, BOOL, INT, FLOAT, STR, SEQ, MAP } kind = NUL;
    bool b = false;
    long long i = 0;
    double f = 0.0;
    std::string s;
    std::vector<Node> seq;
    std::vector<std::pair<std::string, Node>> map;  // insertion order; lookups take the first match

    // serde_yaml 0.9 (the reference pins 0.9.21, parse_scene/Cargo.toml:9) refuses a mapping that names a key twice
    // (its Mapping visitor: "duplicate entry with key ..."), so parse_scene fails with ParseError::Yaml
    void insert(const std::string &key, Node v, int line) {
        for (const auto &kv : map)
            if (kv.first == key)
                fail("YAML deserialization error: duplicate entry with key \"" + key + "\" on line " + std::to_string(line));
        map.emplace_back(key, std::move(v));
    }
    const Node *get(const std::string &key) const {
        if (kind != MAP) return nullptr;
        for (const auto &kv : map)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    bool is_number() const { return kind == INT || kind == FLOAT; }
    double number() const { return kind == INT ? (double)i : f; }  // serde_yaml Value::as_f64
    std::string repr() const {
        switch (kind) {
        case NUL: return "null";
        case BOOL: return b ? "true" : "false";
        case INT: return std::to_string(i);
        case FLOAT: return std::to_string(f);
        case STR: return "\"" + s + "\"";
        case SEQ: return "<sequence>";
        default: return "<mapping>";
        }
    }
};

struct Line {
    int indent;
    std::string text;
    int no;
};

std::string rtrim(std::string s) {
    while (!s.empty() && (s.back() == ' ' || s.back() == '\t' || s.back() == '\r')) s.pop_back();
    return s;
}
std::string trim(const std::string &s) {
    size_t a = 0;
    while (a < s.size() && (s[a] == ' ' || s[a] == '\t')) a++;
    return rtrim(s.substr(a));
}

// removes a trailing comment: '#' at the start of the text or after whitespace, outside quotes
std::string strip_comment(const std::string &s) {
    char quote = 0;
    for (size_t k = 0; k < s.size(); k++) {
        const char c = s[k];
        if (quote) {
            if (c == quote) quote = 0;
        } else if (c == '"' || c == '\'') {
            quote = c;
        } else if (c == '#' && (k == 0 || s[k - 1] == ' ' || s[k - 1] == '\t')) {
            return s.substr(0, k);
        }
    }
    return s;
}

bool match_int(const std::string &s) {
    size_t k = 0;
    if (k < s.size() && (s[k] == '-' || s[k] == '+')) k++;
    if (k == s.size()) return false;
    for (; k < s.size(); k++)
        if (!isdigit((unsigned char)s[k])) return false;
    return true;
}
bool match_float(const std::string &s) {
    // [-+]?(\.[0-9]+|[0-9]+(\.[0-9]*)?)([eE][-+]?[0-9]+)?
    size_t k = 0;
    if (k < s.size() && (s[k] == '-' || s[k] == '+')) k++;
    size_t digits = 0;
    while (k < s.size() && isdigit((unsigned char)s[k])) k++, digits++;
    if (k < s.size() && s[k] == '.') {
        k++;
        size_t frac = 0;
        while (k < s.size() && isdigit((unsigned char)s[k])) k++, frac++;
        if (digits == 0 && frac == 0) return false;
    } else if (digits == 0) {
        return false;
    }
    if (k < s.size() && (s[k] == 'e' || s[k] == 'E')) {
        k++;
        if (k < s.size() && (s[k] == '-' || s[k] == '+')) k++;
        size_t ex = 0;
        while (k < s.size() && isdigit((unsigned char)s[k])) k++, ex++;
        if (ex == 0) return false;
    }
    return k == s.size();
}

// the strings Rust's f64::from_str accepts: [+-]? ( inf | infinity | nan, ASCII case-insensitive | decimal ), decimal =
// digits [. digits*] [exp] | . digits [exp], exp = [eE] [+-]? digits
bool rust_f64_syntax(const std::string &s) {
    size_t k = 0;
    if (k < s.size() && (s[k] == '-' || s[k] == '+')) k++;
    std::string low;
    for (size_t q = k; q < s.size(); q++) low.push_back((char)tolower((unsigned char)s[q]));
    if (low == "inf" || low == "infinity" || low == "nan") return true;
    return k < s.size() && match_float(s.substr(k));
}

// serde_yaml 0.9: an integer spelt with leading zeros is a string (de.rs digits_but_not_number)
bool digits_but_not_number(const std::string &v) {
    const std::string t = (!v.empty() && (v[0] == '-' || v[0] == '+')) ? v.substr(1) : v;
    if (t.size() < 2 || t[0] != '0') return false;
    for (size_t k = 1; k < t.size(); k++)
        if (!isdigit((unsigned char)t[k])) return false;
    return true;
}

// Plain scalars are typed the way serde_yaml 0.9 types them (de.rs visit_untagged_scalar; the reference pins 0.9.21,
// parse_scene/Cargo.toml:9 -- the crate is not vendored, these rules restate its published source: parity unpinned):
// null; true / false; integers (decimal, 0x / 0o / 0b, optional sign first; leading zeros make a string; beyond 64 bits:
// the float rule); floats (.inf / -.inf / .nan, else f64::from_str after an optional "+" if the value is finite); string.
Node scalar(const std::string &raw) {
    Node n;
    const std::string s = trim(raw);
    if (s.empty() || s == "~" || s == "null" || s == "Null" || s == "NULL") return n;
    if (s.size() >= 2 && (s.front() == '"' || s.front() == '\'') && s.back() == s.front()) {
        n.kind = Node::STR;
        n.s = s.substr(1, s.size() - 2);
        return n;
    }
    if (s == "true" || s == "True" || s == "TRUE" || s == "false" || s == "False" || s == "FALSE") {
        n.kind = Node::BOOL;
        n.b = (s[0] == 't' || s[0] == 'T');
        return n;
    }
    {
        const bool neg = s[0] == '-';
        const std::string body = (s[0] == '-' || s[0] == '+') ? s.substr(1) : s;
        if (!body.empty() && body[0] != '-' && body[0] != '+') {
            int radix = 10;
            std::string digits = body;
            if (body.compare(0, 2, "0x") == 0) radix = 16, digits = body.substr(2);
            else if (body.compare(0, 2, "0o") == 0) radix = 8, digits = body.substr(2);
            else if (body.compare(0, 2, "0b") == 0) radix = 2, digits = body.substr(2);
            bool ok = !digits.empty() && (radix != 10 || !digits_but_not_number(s));
            for (char c : digits) {
                const int d = isdigit((unsigned char)c) ? c - '0' : isxdigit((unsigned char)c) ? tolower((unsigned char)c) - 'a' + 10 : 99;
                ok = ok && d < radix;
            }
            if (ok) {
                errno = 0;
                const unsigned long long u = strtoull(digits.c_str(), nullptr, radix);
                // i64 for negative values, u64 for the others (serde_yaml tries u64, then i64); beyond: the float rule
                if (errno == 0 && (!neg || u <= (1ull << 63))) {
                    n.kind = Node::INT;
                    if (neg) {
                        n.i = (long long)(0ull - u);
                    } else if (u <= (unsigned long long)std::numeric_limits<long long>::max()) {
                        n.i = (long long)u;
                    } else {
                        n.kind = Node::FLOAT;  // (a u64 above i64::MAX: Value::as_f64 of it; nothing here reads such an integer)
                        n.f = (double)u;
                    }
                    return n;
                }
            }
        }
    }
    if (!digits_but_not_number(s)) {
        const std::string unpositive = s[0] == '+' ? s.substr(1) : s;
        const bool double_sign = s[0] == '+' && !unpositive.empty() && (unpositive[0] == '-' || unpositive[0] == '+');
        if (!double_sign) {
            if (unpositive == ".inf" || unpositive == ".Inf" || unpositive == ".INF" || s == "-.inf" || s == "-.Inf" || s == "-.INF") {
                n.kind = Node::FLOAT;
                n.f = (s[0] == '-') ? -std::numeric_limits<double>::infinity() : std::numeric_limits<double>::infinity();
                return n;
            }
            if (s == ".nan" || s == ".NaN" || s == ".NAN") {
                n.kind = Node::FLOAT;
                n.f = std::numeric_limits<double>::quiet_NaN();
                return n;
            }
            if (rust_f64_syntax(unpositive)) {
                const double f = strtod(unpositive.c_str(), nullptr);
                if (std::isfinite(f)) {
                    n.kind = Node::FLOAT;
                    n.f = f;
                    return n;
                }
            }
        }
    }
    n.kind = Node::STR;
    n.s = s;
    return n;
}

// position of the ':' that separates key and value in a block mapping line, or npos
size_t key_colon(const std::string &s) {
    char quote = 0;
    int depth = 0;
    for (size_t k = 0; k < s.size(); k++) {
        const char c = s[k];
        if (quote) {
            if (c == quote) quote = 0;
            continue;
        }
        if (c == '"' || c == '\'') quote = c;
        else if (c == '[' || c == '{') depth++;
        else if (c == ']' || c == '}') depth--;
        else if (c == ':' && depth == 0 && (k + 1 == s.size() || s[k + 1] == ' ' || s[k + 1] == '\t')) return k;
    }
    return std::string::npos;
}

struct Flow {
    const std::string &s;
    size_t p = 0;
    int line;
    void ws() {
        while (p < s.size() && (s[p] == ' ' || s[p] == '\t')) p++;
    }
    Node value() {
        ws();
        if (p >= s.size()) fail("YAML deserialization error: unexpected end of flow collection on line " + std::to_string(line));
        if (s[p] == '[') {
            p++;
            Node n;
            n.kind = Node::SEQ;
            for (;;) {
                ws();
                if (p >= s.size()) fail("YAML deserialization error: unterminated '[' on line " + std::to_string(line));
                if (s[p] == ']') {
                    p++;
                    return n;
                }
                n.seq.push_back(value());
                ws();
                if (p < s.size() && s[p] == ',') p++;
            }
        }
        if (s[p] == '{') {
            p++;
            Node n;
            n.kind = Node::MAP;
            for (;;) {
                ws();
                if (p >= s.size()) fail("YAML deserialization error: unterminated '{' on line " + std::to_string(line));
                if (s[p] == '}') {
                    p++;
                    return n;
                }
                const size_t start = p;
                while (p < s.size() && s[p] != ':' && s[p] != ',' && s[p] != '}') p++;
                const std::string key = scalar(s.substr(start, p - start)).s;
                Node v;
                if (p < s.size() && s[p] == ':') {
                    p++;
                    v = value();
                }
                n.insert(key, v, line);
                ws();
                if (p < s.size() && s[p] == ',') p++;
            }
        }
        // plain or quoted scalar up to the next , ] }
        const size_t start = p;
        if (s[p] == '"' || s[p] == '\'') {
            const char q = s[p++];
            while (p < s.size() && s[p] != q) p++;
            if (p < s.size()) p++;
        } else {
            while (p < s.size() && s[p] != ',' && s[p] != ']' && s[p] != '}') p++;
        }
        return scalar(s.substr(start, p - start));
    }
};

Node inline_value(const std::string &text, int line) {
    const std::string s = trim(text);
    if (!s.empty() && (s[0] == '[' || s[0] == '{')) {
        Flow f{s, 0, line};
        Node n = f.value();
        f.ws();
        if (f.p != s.size()) fail("YAML deserialization error: trailing characters after flow collection on line " + std::to_string(line));
        return n;
    }
    return scalar(s);
}

struct Yaml {
    std::vector<Line> lines;

    explicit Yaml(const std::string &text) {
        std::istringstream in(text);
        std::string raw;
        int no = 0;
        while (std::getline(in, raw)) {
            no++;
            std::string s = rtrim(strip_comment(raw));
            size_t a = 0;
            while (a < s.size() && s[a] == ' ') a++;
            if (a < s.size() && s[a] == '\t') fail("YAML deserialization error: tab used for indentation on line " + std::to_string(no));
            if (a == s.size()) continue;
            if (s.compare(a, 3, "---") == 0 && a == 0) continue;
            lines.push_back({(int)a, s.substr(a), no});
        }
    }
    static bool is_item(const std::string &t) { return t[0] == '-' && (t.size() == 1 || t[1] == ' '); }

    Node node(size_t &i, int indent) {
        const Line &l = lines[i];
        if (is_item(l.text)) return sequence(i, indent);
        if (l.text[0] != '[' && l.text[0] != '{' && key_colon(l.text) != std::string::npos) return mapping(i, indent);
        Node n = inline_value(l.text, l.no);
        i++;
        return n;
    }
    Node sequence(size_t &i, int indent) {
        Node n;
        n.kind = Node::SEQ;
        while (i < lines.size() && lines[i].indent == indent && is_item(lines[i].text)) {
            const std::string &t = lines[i].text;
            size_t off = 1;
            while (off < t.size() && t[off] == ' ') off++;
            if (off >= t.size()) {
                i++;
                if (i < lines.size() && lines[i].indent > indent)
                    n.seq.push_back(node(i, lines[i].indent));
                else
                    n.seq.push_back(Node());
            } else {
                // the rest of the line is the first line of the item's own block
                lines[i] = Line{indent + (int)off, t.substr(off), lines[i].no};
                n.seq.push_back(node(i, indent + (int)off));
            }
        }
        if (i < lines.size() && lines[i].indent > indent)
            fail("YAML deserialization error: bad indentation on line " + std::to_string(lines[i].no));
        return n;
    }
    Node mapping(size_t &i, int indent) {
        Node n;
        n.kind = Node::MAP;
        while (i < lines.size() && lines[i].indent == indent && !is_item(lines[i].text)) {
            const Line l = lines[i];
            const size_t c = key_colon(l.text);
            if (c == std::string::npos) fail("YAML deserialization error: expected 'key: value' on line " + std::to_string(l.no));
            const Node key = scalar(l.text.substr(0, c));
            const std::string k = key.kind == Node::STR ? key.s : trim(l.text.substr(0, c));
            const std::string rest = trim(l.text.substr(c + 1));
            Node v;
            if (!rest.empty()) {
                v = inline_value(rest, l.no);
                i++;
            } else {
                i++;
                if (i < lines.size() && lines[i].indent > indent)
                    v = node(i, lines[i].indent);
                else if (i < lines.size() && lines[i].indent == indent && is_item(lines[i].text))
                    v = sequence(i, indent);  // "key:" followed by "- item" at the key's own indentation
            }
            n.insert(k, std::move(v), l.no);
        }
        if (i < lines.size() && lines[i].indent > indent)
            fail("YAML deserialization error: bad indentation on line " + std::to_string(lines[i].no));
        return n;
    }
    Node document() {
        if (lines.empty()) return Node();
        size_t i = 0;
        Node n = node(i, lines[0].indent);
        if (i != lines.size()) fail("YAML deserialization error: unexpected content on line " + std::to_string(lines[i].no));
        return n;
    }
};

// ------------------------------------------------------------------------------------------------
// parse_scene
// ------------------------------------------------------------------------------------------------
double get_f64(const Node &v) {  // YamlHelper::get_f64
    if (!v.is_number()) fail("Invalid value: " + v.repr());
    return v.number();
}
std::vector<double> get_f64_vec(const Node &v) {  // YamlHelper::get_f64_vec
    if (v.kind != Node::SEQ) fail("Invalid value: " + v.repr());
    std::vector<double> r;
    for (const Node &x : v.seq) r.push_back(get_f64(x));
    return r;
}
void need3(const std::vector<double> &v, const char *what) {
    if (v.size() < 3) fail(std::string("Invalid value: ") + what + " needs 3 components");
}
double f64_or(const Node &m, const char *key, double dflt) {
    const Node *v = m.get(key);
    return (v && v->is_number()) ? v->number() : dflt;
}

struct Transform {  // parse_scene/src/lib.rs:94-116
    enum Kind { SCALE, TRANSLATE, ROTATE_X, ROTATE_Y, ROTATE_Z } kind;
    double a = 0, b = 0, c = 0;
};

Transform parse_transformation(const Node &item) {  // parse_scene/src/lib.rs:118-192
    if (item.kind != Node::SEQ) fail("Invalid value: " + item.repr());
    if (item.seq.empty()) fail("Invalid parameters for transformation: Missing transformation name");
    if (item.seq[0].kind != Node::STR) fail("Invalid value: " + item.seq[0].repr());
    const std::string &name = item.seq[0].s;
    Transform t{};
    if (name == "scale" || name == "translate") {
        std::vector<double> p;
        for (size_t k = 1; k < item.seq.size(); k++) p.push_back(get_f64(item.seq[k]));
        if (p.size() < 3) fail("Invalid parameters for transformation: " + name + " needs 3 parameters");
        t.kind = name == "scale" ? Transform::SCALE : Transform::TRANSLATE;
        t.a = p[0], t.b = p[1], t.c = p[2];
        return t;
    }
    if (name == "rotate-x" || name == "rotate-y" || name == "rotate-z") {
        if (item.seq.size() < 2) fail("Invalid parameters for transformation: Missing transformation parameters");
        t.kind = name == "rotate-x" ? Transform::ROTATE_X : name == "rotate-y" ? Transform::ROTATE_Y : Transform::ROTATE_Z;
        t.a = get_f64(item.seq[1]);
        return t;
    }
    fail("Invalid parameters for transformation: Unknown transformation name: " + name);
}

// the Matrix fluent API left-multiplies (matrix.rs:238-266)
Mat4 apply(const Mat4 &m, const Transform &t) {
    switch (t.kind) {
    case Transform::SCALE: return mat_mul(mat_scaling(t.a, t.b, t.c), m);
    case Transform::TRANSLATE: return mat_mul(mat_translation(t.a, t.b, t.c), m);
    case Transform::ROTATE_X: return mat_mul(mat_rotation_x(t.a), m);
    case Transform::ROTATE_Y: return mat_mul(mat_rotation_y(t.a), m);
    default: return mat_mul(mat_rotation_z(t.a), m);
    }
}

rtc_material parse_material(const Node &desc) {  // parse_scene/src/lib.rs:488-520
    rtc_material m;
    rtc_material_default(&m);
    static const Node empty;
    const Node &d = desc.kind == Node::MAP ? desc : empty;
    if (const Node *c = d.get("color")) {
        std::vector<double> cv{1., 1., 1.};
        bool ok = c->kind == Node::SEQ;
        if (ok)
            for (const Node &x : c->seq) ok = ok && x.is_number();
        if (ok) cv = get_f64_vec(*c);  // .and_then(|x| x.get_f64_vec().ok()).unwrap_or_else(|| vec![1., 1., 1.])
        need3(cv, "color");
        m.color[0] = cv[0], m.color[1] = cv[1], m.color[2] = cv[2];
    }
    m.ambient = f64_or(d, "ambient", 0.1);
    m.diffuse = f64_or(d, "diffuse", 0.9);
    m.specular = f64_or(d, "specular", 0.9);
    m.shininess = f64_or(d, "shininess", 200.);
    m.reflectivity = f64_or(d, "reflective", 0.);
    m.transparency = f64_or(d, "transparency", 0.);
    m.refractive_index = f64_or(d, "refractive-index", 1.);
    const Node *cs = d.get("casts-shadow");
    m.casts_shadow = (cs && cs->kind == Node::BOOL) ? (cs->b ? 1 : 0) : 1;
    return m;
}

struct Definition {
    bool is_material = false;
    rtc_material material{};
    std::vector<Transform> transforms;
};
typedef std::vector<std::pair<std::string, Definition>> Definitions;  // find() returns the FIRST match

const Definition *find_def(const Definitions &defs, const std::string &name) {
    for (const auto &d : defs)
        if (d.first == name) return &d.second;
    return nullptr;
}

Definition parse_definition(const Node &desc, const Definitions &defs) {  // parse_scene/src/lib.rs:194-342
    const Node *value = desc.get("value");
    if (!value) fail("Invalid parameters for a constant definition: Missing 'value' field");
    Definition out;
    if (value->kind == Node::MAP) {
        out.is_material = true;
        const rtc_material nm = parse_material(*value);
        if (const Node *ext = desc.get("extend")) {
            if (ext->kind != Node::STR) fail("Invalid value: " + ext->repr());
            const Definition *orig = find_def(defs, ext->s);
            if (!orig) fail("Invalid parameters for a constant definition: Definition " + ext->s + " not found");
            if (!orig->is_material) fail("Cannot extend a definition that does not have a material");
            rtc_material m = orig->material;
            // Q16: only keys present in `value` override the extended material
            if (value->get("color")) memcpy(m.color, nm.color, sizeof m.color);
            if (value->get("ambient")) m.ambient = nm.ambient;
            if (value->get("diffuse")) m.diffuse = nm.diffuse;
            if (value->get("specular")) m.specular = nm.specular;
            if (value->get("shininess")) m.shininess = nm.shininess;
            if (value->get("reflective")) m.reflectivity = nm.reflectivity;
            if (value->get("refractive-index")) m.refractive_index = nm.refractive_index;
            if (value->get("transparency")) m.transparency = nm.transparency;
            if (value->get("casts-shadow")) m.casts_shadow = nm.casts_shadow;
            out.material = m;
        } else {
            out.material = nm;
        }
        return out;
    }
    if (value->kind == Node::SEQ) {
        for (const Node &item : value->seq) {
            if (item.kind == Node::STR) {
                const Definition *orig = find_def(defs, item.s);
                if (!orig) fail("Invalid parameters for a constant definition: Definition " + item.s + " not found");
                if (orig->is_material) fail("Cannot extend a definition that does not have transformations");
                out.transforms.insert(out.transforms.end(), orig->transforms.begin(), orig->transforms.end());
            } else {
                out.transforms.push_back(parse_transformation(item));
            }
        }
        return out;
    }
    fail("Definition 'value' must be a mapping or a sequence");
}

#define RC(call)                                     \
    do {                                             \
        const int rc__ = (call);                     \
        if (rc__ != RTC_OK) throw rc__;              \
    } while (0)

std::string resolve_path(const std::string &fn, const std::string &base_dir) {
    // The reference opens the path relative to the CWD (samples/teapot.yaml:34 expects the repo root); this build
    // also looks next to the YAML file and one directory above it.
    auto exists = [](const std::string &p) {
        std::ifstream f(p);
        return f.good();
    };
    if (fn.empty() || fn[0] == '/' || exists(fn) || base_dir.empty()) return fn;
    std::string cand = base_dir + "/" + fn;
    if (exists(cand)) return cand;
    const size_t slash = base_dir.find_last_of('/');
    if (slash != std::string::npos) {
        cand = base_dir.substr(0, slash) + "/" + fn;
        if (exists(cand)) return cand;
    }
    return fn;
}

struct ObjResult {
    uint32_t ignored_lines = 0, vertices = 0, normals = 0, triangles = 0, groups = 0;
};

// obj_file/src/lib.rs:40-230.  The default group carries `transform` and `material`; every `g` record opens a
// named child group (identity transform, same material) that takes all following faces.
void add_obj(rtc_scene *scene, const std::string &text, const Mat4 &transform, int32_t material_id, ObjResult &res) {
    double t16[16];
    transform.to16(t16);
    RC(rtc_scene_begin_group(scene, t16, material_id));
    uint32_t tri_material = 0;
    {
        rtc_material dm;
        rtc_material_default(&dm);  // Triangle::new(.., Material::default()): the parent group's material wins
        RC(rtc_scene_add_material(scene, &dm, &tri_material));
    }
    std::vector<Vec3> vertices, normals;
    std::vector<double> run_pts, run_nrm;
    bool run_smooth = false, in_named = false;
    auto flush = [&]() {
        if (run_pts.empty()) return;
        RC(rtc_scene_add_triangles(scene, run_pts.size() / 9, run_pts.data(), run_smooth ? run_nrm.data() : nullptr,
                                   tri_material));
        res.triangles += (uint32_t)(run_pts.size() / 9);
        run_pts.clear(), run_nrm.clear();
    };
    std::istringstream in(text);
    std::string line;
    int n = -1;
    while (std::getline(in, line)) {
        n++;
        const int line_no = n + 1;
        std::vector<std::string> tok;
        {
            std::istringstream ls(line);
            std::string w;
            while (ls >> w) tok.push_back(w);
        }
        if (tok.empty()) continue;
        if (tok[0] == "v" || tok[0] == "vn") {
            if (tok.size() < 4) continue;  // "Malformed vertex record" is printed and the line skipped (:122-124)
            double c[3];
            for (int k = 0; k < 3; k++) {
                // f64::from_str (obj_file/src/lib.rs:82-96 `.parse::<f64>()`): ASCII decimal syntax or inf / infinity /
                // nan in any case -- strtod alone would also take hexadecimal floats and "nan(...)"
                if (!rust_f64_syntax(tok[1 + k]))
                    fail("invalid input: Invalid vertex record on line " + std::to_string(line_no) + ": " + tok[1 + k]);
                c[k] = strtod(tok[1 + k].c_str(), nullptr);
            }
            (tok[0] == "v" ? vertices : normals).push_back({c[0], c[1], c[2]});
        } else if (tok[0] == "f") {
            std::vector<Vec3> face;
            std::vector<size_t> nidx;
            auto parse_index = [&](const std::string &s, const char *what) -> size_t {
                size_t k = (!s.empty() && s[0] == '+') ? 1 : 0;
                if (k == s.size()) k = std::string::npos;
                for (size_t q = k; q < s.size(); q++)
                    if (!isdigit((unsigned char)s[q])) k = std::string::npos;
                if (k == std::string::npos)
                    fail(std::string("invalid input: Invalid ") + what + " index for face record on line " +
                         std::to_string(n) + ": " + s);
                return (size_t)strtoull(s.c_str(), nullptr, 10);
            };
            for (size_t k = 1; k < tok.size(); k++) {
                std::vector<std::string> sub;
                {
                    size_t a = 0;
                    for (;;) {
                        const size_t b = tok[k].find('/', a);
                        sub.push_back(tok[k].substr(a, b == std::string::npos ? b : b - a));
                        if (b == std::string::npos) break;
                        a = b + 1;
                    }
                }
                const size_t vi = parse_index(sub[0], "vertex");
                if (vi < 1 || vi > vertices.size()) fail("missing vertex at index " + std::to_string(vi));
                face.push_back(vertices[vi - 1]);
                if (sub.size() > 2) {
                    const size_t ni = parse_index(sub[2], "normal");
                    if (ni < 1) fail("invalid input: normal index 0 on line " + std::to_string(line_no));
                    nidx.push_back(ni - 1);
                }
            }
            if (face.size() < 2) fail("invalid input: face with " + std::to_string(face.size()) + " vertices on line " + std::to_string(line_no));
            const bool smooth = !nidx.empty();
            for (size_t k = 1; k + 1 < face.size(); k++) {
                if (smooth != run_smooth) flush();
                run_smooth = smooth;
                const Vec3 p[3] = {face[0], face[k], face[k + 1]};
                for (const Vec3 &v : p) run_pts.push_back(v.x), run_pts.push_back(v.y), run_pts.push_back(v.z);
                if (smooth) {
                    // Q11: every triangle of the fan takes the face's normals 0, 1, 2 (obj_file/src/lib.rs:169-171)
                    if (nidx.size() < 3 || nidx[0] >= normals.size() || nidx[1] >= normals.size() || nidx[2] >= normals.size())
                        fail("invalid input: missing normal on line " + std::to_string(line_no));
                    for (int q = 0; q < 3; q++) {
                        const Vec3 &v = normals[nidx[q]];
                        run_nrm.push_back(v.x), run_nrm.push_back(v.y), run_nrm.push_back(v.z);
                    }
                }
            }
        } else if (tok[0] == "g") {
            if (tok.size() < 2)
                fail("invalid input: Malformed named group record on line " + std::to_string(n) + ": missing group name");
            flush();
            if (in_named) RC(rtc_scene_end_group(scene));
            double id16[16];
            Mat4::identity().to16(id16);
            RC(rtc_scene_begin_group(scene, id16, material_id));
            in_named = true;
            res.groups++;
        } else {
            res.ignored_lines++;
        }
    }
    flush();
    if (in_named) RC(rtc_scene_end_group(scene));
    RC(rtc_scene_end_group(scene));
    res.vertices = (uint32_t)vertices.size();
    res.normals = (uint32_t)normals.size();
}

std::string read_file(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw(int) RTC_ERR_IO;
    std::ostringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

struct Loader {
    rtc_scene *scene;
    std::string base_dir;
    Definitions defs;

    uint32_t add_material(const rtc_material &m) {
        uint32_t id = 0;
        RC(rtc_scene_add_material(scene, &m, &id));
        return id;
    }

    void construct_object(const Node &d) {  // parse_scene/src/lib.rs:522-733
        const Node *add = d.get("add");
        if (!add || add->kind != Node::STR) fail("Invalid value: Object's 'add' must be a string");
        const std::string &kind = add->s;
        rtc_material material;
        rtc_material_default(&material);
        if (const Node *md = d.get("material")) {
            if (md->kind == Node::STR) {
                const Definition *def = find_def(defs, md->s);
                if (!def) fail("Invalid value: No such defintion: " + md->s);
                if (!def->is_material) fail("Invalid value: Definition '" + md->s + "' does not refer to a material");
                material = def->material;
            } else {
                material = parse_material(*md);
            }
        }
        Mat4 t = Mat4::identity();
        const Node *trs = d.get("transform");
        if (trs && trs->kind == Node::SEQ) {
            for (const Node &v : trs->seq) {
                if (v.kind == Node::STR) {
                    const Definition *def = find_def(defs, v.s);
                    if (!def) fail("Invalid value: No such defintion: " + v.s);
                    if (def->is_material)
                        fail("Invalid value: Definition '" + v.s + "' does not refer to a set of transformations");
                    for (const Transform &tr : def->transforms) t = apply(t, tr);
                } else {
                    t = apply(t, parse_transformation(v));
                }
            }
        }
        double t16[16];
        t.to16(t16);
        const double inf = std::numeric_limits<double>::infinity();
        if (kind == "sphere" || kind == "plane" || kind == "cube") {
            const int type = kind == "sphere" ? RTC_SPHERE : kind == "plane" ? RTC_PLANE : RTC_CUBE;
            RC(rtc_scene_add_shape(scene, type, t16, add_material(material), -inf, inf, 0));
        } else if (kind == "cylinder" || kind == "cone") {
            const double minimum = f64_or(d, "minimum", -inf), maximum = f64_or(d, "maximum", inf);
            const Node *c = d.get("closed");
            const int closed = (c && c->kind == Node::BOOL && c->b) ? 1 : 0;
            RC(rtc_scene_add_shape(scene, kind == "cylinder" ? RTC_CYLINDER : RTC_CONE, t16, add_material(material),
                                   minimum, maximum, closed));
        } else if (kind == "group") {
            // Q14: the parsed `material` is dropped; groups carry Some(Material::default()) (shape.rs:1909)
            const Node *children = d.get("children");
            if (!children) fail("Invalid parameters for group: Missing 'children' in group definition");
            if (children->kind != Node::SEQ) fail("Invalid parameters for group: Group 'children' must be a sequence");
            for (const Node &c : children->seq)
                if (c.kind != Node::MAP)
                    fail("Invalid parameters for group: Each value in a group's 'children' must be a mapping");
            rtc_material dm;
            rtc_material_default(&dm);
            RC(rtc_scene_begin_group(scene, t16, (int32_t)add_material(dm)));
            // A nested YAML group is added to its parent as soon as it is constructed (lib.rs:677-679); every other
            // child only after the whole `children` list has been built (lib.rs:707-712): nested groups therefore
            // come first in Group::children order.
            for (int pass = 0; pass < 2; pass++)
                for (const Node &c : children->seq) {
                    const Node *ca = c.get("add");
                    const bool nested = ca && ca->kind == Node::STR && ca->s == "group";
                    if (nested == (pass == 0)) construct_object(c);
                }
            RC(rtc_scene_end_group(scene));
        } else if (kind == "obj") {
            const Node *fn = d.get("file");
            if (!fn) fail("Invalid parameters for obj: Missing 'file' field");
            if (fn->kind != Node::STR) fail("Invalid value: obj 'file' must be a string");
            std::string text;
            try {
                text = read_file(resolve_path(fn->s, base_dir));
            } catch (int) {
                fail("failed to open the file: " + fn->s);
            }
            ObjResult r;
            add_obj(scene, text, t, (int32_t)add_material(material), r);
        } else {
            fail("unimplemented object type: " + kind);  // parse_scene/src/lib.rs:729 unimplemented!()
        }
    }

    void load(const std::string &text, uint32_t width, uint32_t height) {  // parse_scene/src/lib.rs:735-784
        Yaml y(text);
        const Node doc = y.document();
        if (doc.kind != Node::SEQ) fail("YAML deserialization error: expected a sequence");
        bool have_camera = false;
        for (const Node &item : doc.seq) {
            if (item.kind != Node::MAP) fail("Invalid YAML object: Expected mapping, found " + item.repr());
            const Node *add = item.get("add");
            const Node *def = item.get("define");
            if (add && add->kind == Node::STR) {
                const std::string &a = add->s;
                if (a == "camera") {  // parse_scene/src/lib.rs:344-397
                    const Node *w = item.get("width"), *h = item.get("height");
                    if (!w || w->kind != Node::INT || w->i < 0)
                        fail("Invalid parameters for camera: Missing camera width / must be an integer");
                    if (!h || h->kind != Node::INT || h->i < 0)
                        fail("Invalid parameters for camera: Missing camera height / must be an integer");
                    for (const char *key : {"field-of-view", "from", "to", "up"})
                        if (!item.get(key)) fail(std::string("Invalid parameters for camera: Missing '") + key + "'");
                    const double fov = get_f64(*item.get("field-of-view"));
                    const std::vector<double> from = get_f64_vec(*item.get("from")), to = get_f64_vec(*item.get("to")),
                                              up = get_f64_vec(*item.get("up"));
                    need3(from, "from"), need3(to, "to"), need3(up, "up");
                    double v16[16];
                    mat_view({from[0], from[1], from[2]}, {to[0], to[1], to[2]}, {up[0], up[1], up[2]}).to16(v16);
                    // width / height overrides are this build's addition (BASELINE configs resize the samples)
                    RC(rtc_scene_set_camera(scene, width ? width : (uint32_t)w->i, height ? height : (uint32_t)h->i, fov, v16));
                    have_camera = true;
                } else if (a == "light") {  // parse_scene/src/lib.rs:399-431
                    if (!item.get("at")) fail("Invalid parameters for light: Missing 'at'");
                    if (!item.get("intensity")) fail("Invalid parameters for light: Missing 'intensity'");
                    const std::vector<double> at = get_f64_vec(*item.get("at")), in = get_f64_vec(*item.get("intensity"));
                    need3(at, "at"), need3(in, "intensity");
                    RC(rtc_scene_add_point_light(scene, at.data(), in.data()));
                } else if (a == "area-light") {  // parse_scene/src/lib.rs:433-486
                    for (const char *key : {"corner", "full_uvec", "usteps", "full_vvec", "vsteps", "intensity"})
                        if (!item.get(key)) fail(std::string("Invalid parameters for light: Missing '") + key + "'");
                    const Node *us = item.get("usteps"), *vs = item.get("vsteps");
                    if (us->kind != Node::INT || us->i < 0 || vs->kind != Node::INT || vs->i < 0)
                        fail("Invalid parameters for light: Invalid value for usteps/vsteps");
                    const std::vector<double> corner = get_f64_vec(*item.get("corner")), fu = get_f64_vec(*item.get("full_uvec")),
                                              fv = get_f64_vec(*item.get("full_vvec")), in = get_f64_vec(*item.get("intensity"));
                    need3(corner, "corner"), need3(fu, "full_uvec"), need3(fv, "full_vvec"), need3(in, "intensity");
                    RC(rtc_scene_add_area_light(scene, corner.data(), fu.data(), (uint32_t)us->i, fv.data(), (uint32_t)vs->i,
                                                in.data()));
                } else if (a == "sphere" || a == "plane" || a == "cube" || a == "cylinder" || a == "cone" || a == "group" ||
                           a == "obj") {
                    construct_object(item);
                } else {
                    fail("Encountered unknown item: " + a);
                }
            } else if (def && def->kind == Node::STR) {
                Definition d = parse_definition(item, defs);
                defs.emplace_back(def->s, std::move(d));  // lookups return the first definition of a name
            }
        }
        if (!have_camera) fail("The scene is missing camera defintion");
    }
};

int guarded(const std::function<void()> &body) {
    try {
        body();
    } catch (const ParseFail &e) {
        set_error("%s", e.msg.c_str());
        return RTC_ERR_PARSE;
    } catch (int rc) {
        if (rc == RTC_ERR_IO) set_error("file could not be read");
        return rc;  // an rtc_scene_* call failed; its message is already set
    } catch (const std::bad_alloc &) {
        set_error("out of memory");
        return RTC_ERR_INVALID;
    }
    return RTC_OK;
}
}  // namespace

extern "C" {

int rtc_scene_load_yaml(const char *yaml_text, const char *base_dir, uint32_t width, uint32_t height, rtc_scene **out) {
    if (!yaml_text || !out) {
        set_error("yaml_text / out is NULL");
        return RTC_ERR_INVALID;
    }
    *out = nullptr;
    rtc_scene *scene = rtc_scene_create();
    if (!scene) {
        set_error("rtc_scene_create failed");
        return RTC_ERR_INVALID;
    }
    const int rc = guarded([&]() {
        Loader l{scene, base_dir ? base_dir : "", {}};
        l.load(yaml_text, width, height);
    });
    if (rc != RTC_OK) {
        rtc_scene_destroy(scene);
        return rc;
    }
    *out = scene;
    return RTC_OK;
}

int rtc_scene_load_yaml_file(const char *path, uint32_t width, uint32_t height, rtc_scene **out) {
    if (!path || !out) {
        set_error("path / out is NULL");
        return RTC_ERR_INVALID;
    }
    *out = nullptr;
    std::string text;
    try {
        text = read_file(path);
    } catch (int) {
        set_error("cannot open scene file %s", path);
        return RTC_ERR_IO;
    }
    std::string dir(path);
    const size_t slash = dir.find_last_of('/');
    dir = slash == std::string::npos ? "." : dir.substr(0, slash);
    return rtc_scene_load_yaml(text.c_str(), dir.c_str(), width, height, out);
}

int rtc_scene_add_obj(rtc_scene *scene, const char *obj_text, const double transform[16], int32_t material_id,
                      rtc_obj_info *info) {
    if (!scene || !obj_text) {
        set_error("scene / obj_text is NULL");
        return RTC_ERR_INVALID;
    }
    ObjResult r;
    const int rc = guarded([&]() {
        int32_t mid = material_id;
        if (mid < 0) {
            rtc_material dm;
            rtc_material_default(&dm);  // parse_obj_string(.., None, ..): material.unwrap_or_default()
            uint32_t id = 0;
            RC(rtc_scene_add_material(scene, &dm, &id));
            mid = (int32_t)id;
        }
        add_obj(scene, obj_text, transform ? Mat4::from16(transform) : Mat4::identity(), mid, r);
    });
    if (info) {
        info->ignored_lines = r.ignored_lines, info->vertices = r.vertices, info->normals = r.normals;
        info->triangles = r.triangles, info->groups = r.groups;
    }
    return rc;
}

// Canvas::to_ppm, canvas.rs:44-82 with scale_and_clamp, canvas.rs:86-89
int rtc_canvas_to_ppm(const double *rgb, uint32_t hsize, uint32_t vsize, char *out, size_t cap, size_t *needed) {
    if (!rgb && (size_t)hsize * vsize) {
        set_error("rgb is NULL");
        return RTC_ERR_INVALID;
    }
    std::string s = "P3\n" + std::to_string(hsize) + " " + std::to_string(vsize) + "\n255\n";
    s.reserve(s.size() + (size_t)hsize * vsize * 12);
    for (uint32_t y = 0; y < vsize; y++) {
        int chars = 0;
        bool line_start = true;  // nothing pushed yet, or the last thing pushed was a line break
        const double *row = rgb + (size_t)y * hsize * 3;
        for (size_t k = 0; k < (size_t)hsize * 3; k++) {
            if (chars + 4 > 70) {
                s += '\n';
                chars = 0;
                line_start = true;
            }
            if (!line_start) {
                s += ' ';
                chars += 1;
            }
            // (val * 255.).round() as u8: round half away from zero; `as u8` saturates, NaN -> 0
            const double v = std::round(row[k] * 255.0);
            const int q = (v != v) ? 0 : v <= 0.0 ? 0 : v >= 255.0 ? 255 : (int)v;
            s += std::to_string(q);
            chars += 3;  // "3 chars for each subpixel", whatever its printed width
            line_start = false;
        }
        s += '\n';
    }
    if (needed) *needed = s.size();
    if (out && cap >= s.size()) {
        memcpy(out, s.data(), s.size());
        if (cap > s.size()) out[s.size()] = '\0';
        return RTC_OK;
    }
    if (!out) return RTC_OK;  // size query
    set_error("PPM needs %zu bytes, the buffer holds %zu", s.size(), cap);
    return RTC_ERR_INVALID;
}

}  // extern "C"
