// Configuration surface of the tools: `exe [key=value ...] [file.cfg]`.
//
// Host-only part of the drop-in boundary: the reference's executables take their whole
// configuration through DataFile(argc, argv) -> DataFileEngineCfg (src/tools/DataFileEngineCfg.cpp:61-80)
// -> the `rc_*` functions of src/tools/readconfig.c.  This file provides the same grammar and the same typed
// look-ups behind the C ABI (`ecckd_cfg_*`, include/ecckd_hip.h), written over std::string/std::vector:
//
//   param value            '#' comments, values in '...', "..." (may span lines) or {...} (comments stripped)
//   \begin sec ... \end    stored as sec.param, nested sections joined with '.'
//   \include file          relative to the directory of the including file
//   $name                  a value that starts with '$' is replaced by the value of `name`
//   name[m] / name[m][n]   declared vector / matrix dimensions
//   (a b c) v v v v ...    table: values dealt round-robin to the listed names
//   look-ups are case-insensitive; "scope" look-ups match `scope.param`
//
// Behaviours of the reference that look accidental are kept because configurations in the wild rely on what
// the parser does, not on what it meant (each is pinned against the reference's own readconfig.c compiled
// into oracle/_ref by tests/test_config.py):
//   * an empty value ("" or a bare parameter) reads as "1" (readconfig.c:52-55);
//   * inside a section a repeated parameter is appended, not replaced (:486-493 compares ".name" with "name");
//   * when an existing parameter is replaced its declared dimensions are not updated;
//   * the first hyphen argument that is not "--" switches hyphen skipping off in the search for the
//     configuration file (:1063-1068), and only names containing ".cfg" qualify (:1077);
//   * `a=b=c` registers both a -> "b=c" and "a=b" -> "c" (:916-947);
//   * a section opened before any parameter exists is closed by its first parameter (:519).
// One reference behaviour is NOT kept: `key=$missing` on the command line spins for ever there (:926-930,
// `continue` without advancing); here the argument is skipped.
#include <strings.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "common.hpp"

namespace {

struct Entry {
  std::string param;
  std::string value;
  bool has_value = false;   // table columns that received no value stay without one
  int m = 0, n = 0;
};

inline bool is_space(int c) { return c <= ' '; }

std::string compress_whitespace(const std::string& s) {   // readconfig.c:104-134
  std::string out;
  bool in_word = false, space_required = false;
  for (unsigned char ch : s) {
    if (ch > ' ') {
      if (!in_word && space_required) out.push_back(' ');
      space_required = false;
      out.push_back((char)ch);
      in_word = true;
    } else if (in_word) {
      space_required = true;
      in_word = false;
    }
  }
  return out;
}

void strip_trailing(std::string& s) {
  while (!s.empty() && (unsigned char)s.back() <= ' ') s.pop_back();
}

// [m] / [m][n] after a table column name (readconfig.c:215-243)
void parse_table_dims(const char*& c, int& m, int& n) {
  c++;   // past '['
  if (*c) {
    char* e = nullptr;
    long v = std::strtol(c, &e, 10);
    if (v > 0) m = (int)v;
    if (e && e > c) {
      c = e;
      if (*c == ']') {
        c++;
        if (*c == '[') {
          c++;
          if (*c) {
            long w = std::strtol(c, &e, 10);
            if (w > 0) n = (int)w;
          }
        }
      }
    }
  }
  while (*c && (unsigned char)*c > ' ') c++;
}

// number of whitespace-separated items, {..} and ".." counting as one (readconfig.c:1571-1617)
int count_substrings(const std::string& s) {
  const char* c = s.c_str();
  int count = 0;
  while (*c) {
    while (*c && (unsigned char)*c <= ' ') c++;
    if (!*c) return 0;            // trailing whitespace voids the count, as in the reference
    if (*c == '{') { c++; while (*c && *c != '}') c++; if (*c) c++; }
    else if (*c == '"') { c++; while (*c && *c != '"') c++; if (*c) c++; }
    else while ((unsigned char)*c > ' ') c++;
    count++;
  }
  return count;
}

bool substring(const std::string& s, int i, std::string& out) {   // readconfig.c:1502-1566
  const char* c = s.c_str();
  int count = 0;
  while (*c) {
    while (*c && (unsigned char)*c <= ' ') c++;
    if (!*c) return false;
    const char *start, *end;
    if (*c == '{') { c++; start = c; while (*c && *c != '}') c++; end = c; if (*c) c++; }
    else if (*c == '"') { c++; start = c; while (*c && *c != '"') c++; end = c; if (*c) c++; }
    else { start = c; while ((unsigned char)*c > ' ') c++; end = c; }
    if (count == i) { out.assign(start, end); return true; }
    count++;
  }
  return false;
}

}  // namespace

struct ecckd_cfg {
  std::vector<Entry> entries;
  std::string section;       // section in force (while parsing, or set for a scoped look-up)
  bool have_section = false;
  std::string file_name;     // configuration file taken from the command line, "" if none

  const Entry* find_plain(const char* param) const {
    for (const Entry& e : entries)
      if (strcasecmp(param, e.param.c_str()) == 0) return &e;
    return nullptr;
  }
  Entry* find_plain(const char* param) { return const_cast<Entry*>(static_cast<const ecckd_cfg*>(this)->find_plain(param)); }

  // readconfig.c:153-178
  const Entry* find(const char* scope, const char* param) const {
    if (!scope) return find_plain(param);
    const size_t len = std::strlen(scope);
    for (const Entry& e : entries) {
      if (strncasecmp(scope, e.param.c_str(), len) == 0 && e.param.size() > len && e.param[len] == '.' &&
          strcasecmp(e.param.c_str() + len + 1, param) == 0)
        return &e;
    }
    return nullptr;
  }

  // readconfig.c:418-531.  `value` == nullptr: no value given.
  void reg(std::string param, const std::string* value) {
    int m = 0, n = 0;
    const size_t br = param.find('[');
    if (br != std::string::npos) {
      const std::string dims = param.substr(br + 1);
      param.resize(br);
      if (br > 0 && !dims.empty()) {
        const char* c = dims.c_str();
        char* e = nullptr;
        long v = std::strtol(c, &e, 10);
        if (v > 0) m = (int)v;
        if (e && e > c) {
          c = e;
          while (*c && *c != '[') c++;
          if (*c) {             // a second '[' (the reference reads past the terminator when there is none)
            c++;
            if (*c) {
              long w = std::strtol(c, &e, 10);
              if (w > 0) n = (int)w;
            }
          }
        }
      }
    }
    std::string val;
    bool have = value != nullptr;
    if (have) val = *value;
    if (have && !val.empty() && val[0] == '$') {
      if (const Entry* s = find_plain(val.c_str() + 1)) val = s->has_value ? s->value : std::string("1");
    }
    if (!have) val = "1";        // REPLACE_VALUE: a missing value reads as "1"
    for (Entry& e : entries) {
      if (have_section) {
        const size_t len = section.size();
        if (strncasecmp(section.c_str(), e.param.c_str(), len) == 0 && e.param.size() > len && e.param[len] == '.' &&
            strcasecmp(e.param.c_str() + len, param.c_str()) == 0) {     // sic: ".name" against "name"
          e.value = val; e.has_value = true;
          return;
        }
      } else if (strcasecmp(param.c_str(), e.param.c_str()) == 0) {
        e.value = val; e.has_value = true;
        return;
      }
    }
    Entry e;
    e.param = have_section ? section + "." + param : param;
    e.value = val; e.has_value = true;
    e.m = m; e.n = n;
    // The reference keeps the section in force in the list head, which is also the node the very first
    // parameter is written to (:519 clears it): a section opened before anything was defined ends with its
    // first parameter.
    if (entries.empty()) { section.clear(); have_section = false; }
    entries.push_back(std::move(e));
  }

  // readconfig.c:180-408
  bool reg_table(const std::string& params, const std::string& value) {
    std::vector<size_t> cols;
    const char* p = params.c_str();
    while (*p) {
      while (*p && (unsigned char)*p <= ' ') p++;
      if (!*p) break;
      const char* c = p;
      while ((unsigned char)*c > ' ' && *c != '[') c++;
      std::string name(p, c);
      int m = 0, n = 0;
      if (*c == '[') parse_table_dims(c, m, n);
      if (have_section) name = section + "." + name;
      size_t idx = entries.size();
      for (size_t i = 0; i < entries.size(); ++i)
        if (strcasecmp(name.c_str(), entries[i].param.c_str()) == 0) { idx = i; break; }
      if (idx == entries.size()) {
        Entry e; e.param = name;
        entries.push_back(std::move(e));
      }
      entries[idx].value.clear(); entries[idx].has_value = false;
      entries[idx].m = m; entries[idx].n = n;
      cols.push_back(idx);
      p = c;
    }
    if (cols.empty()) return false;
    const char* v = value.c_str();
    size_t k = 0;
    while (*v) {
      while (*v && (unsigned char)*v <= ' ') v++;
      if (!*v) break;
      const char* c = v;
      if (*c == '{') { c++; while (*c && *c != '}') c++; if (*c) c++; }
      else if (*c == '"') { c++; while (*c && *c != '"') c++; if (*c) c++; }
      else while ((unsigned char)*c > ' ') c++;
      Entry& e = entries[cols[k]];
      e.value.push_back(' ');
      e.value.append(v, c);
      e.has_value = true;
      if (++k >= cols.size()) k = 0;
      v = c;
    }
    return true;
  }

  // readconfig.c:557-880
  int append_file(const std::string& path) {
    FILE* f = std::fopen(path.c_str(), "r");
    ECCKD_REQUIRE(f, "Error opening %s", path.c_str());
    std::string text;
    char buf[65536];
    size_t got;
    while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, got);
    std::fclose(f);
    return append_text(text, path);
  }

  int append_text(const std::string& text, const std::string& path) {
    size_t pos = 0;
    const size_t len = text.size();
    auto get = [&]() -> int { return pos < len ? (unsigned char)text[pos++] : (pos++, EOF); };
    auto unget = [&]() { pos--; };
    auto skip_line = [&]() { int c; do { c = get(); } while (c != '\n' && c != EOF); };
    auto skip_ws = [&]() -> int { int c; do { c = get(); } while (c <= ' ' && c != '\n' && c != EOF); return c; };
    for (;;) {
      int c = skip_ws();
      if (c == EOF) break;
      if (c == '#') { skip_line(); continue; }
      if (c == '\n') continue;
      std::string param, value;
      bool have_value = false;
      if (c == '(') {
        while (c != ')') {
          ECCKD_REQUIRE(c != EOF, "%s: file ended before table column names finished: \"%s\"", path.c_str(), param.c_str());
          param.push_back((char)c);
          c = get();
        }
      } else {
        while (c > ' ' && c != '#' && c != EOF) { param.push_back((char)c); c = get(); }
        unget();
      }
      c = skip_ws();
      if (c == '#') {
        skip_line();
      } else if (c != '\n') {
        if (c == '\'' || c == '"') {
          const int quote = c;
          c = get();
          while (c != EOF && c != quote) { value.push_back((char)c); have_value = true; c = get(); }
        } else if (c == '{') {
          c = get();
          while (c != EOF && c != '}') {
            if (c == '#') skip_line();
            else { value.push_back((char)c); have_value = true; }
            c = get();
          }
        } else {
          while (c != EOF && c != '\n') {
            if (c == '#') { skip_line(); break; }
            if (c != '\r') { value.push_back((char)c); have_value = true; }
            c = get();
          }
          strip_trailing(value);
        }
      }
      if (param[0] == '\\') {
        if (strcasecmp(param.c_str(), "\\begin") == 0) {
          if (have_section) {
            section += "." + value;
          } else if (have_value) {
            section = value; have_section = true;
          }
        } else if (strcasecmp(param.c_str(), "\\end") == 0) {
          ECCKD_REQUIRE(have_section, "%s: \"\\end\" with no \"\\begin\"", path.c_str());
          const size_t dot = section.rfind('.');
          const std::string last = dot == std::string::npos ? section : section.substr(dot + 1);
          if (have_value)
            ECCKD_REQUIRE(strcasecmp(last.c_str(), value.c_str()) == 0, "%s: \"\\begin %s\" ended by \"\\end %s\"", path.c_str(),
                          last.c_str(), value.c_str());
          if (dot == std::string::npos) { section.clear(); have_section = false; }
          else section.resize(dot);
        } else if (strcasecmp(param.c_str(), "\\include") == 0) {
          ECCKD_REQUIRE(have_value, "%s: \\include does not specify a file", path.c_str());
          std::string inc = value;
          if (inc[0] != '/') {
            const size_t slash = path.rfind('/');
            if (slash != std::string::npos) inc = path.substr(0, slash + 1) + value;
          }
          ECCKD_REQUIRE(inc != path, "%s attempts to \\include itself", path.c_str());
          ECCKD_CHECK(append_file(inc));
        }   // any other command is ignored
      } else if (param[0] == '(') {
        ECCKD_REQUIRE(have_value && reg_table(param.substr(1), value), "%s: error assigning table of values", path.c_str());
      } else {
        reg(param, have_value ? &value : nullptr);
      }
    }
    return ECCKD_OK;
  }

  // rc_register_files (readconfig.c:958-1005)
  void register_files(int argc, const char* const* argv) {
    std::string v = argv[0];
    reg("0", &v);
    int nfiles = 1;
    for (int i = 1; i < argc; ++i) {
      if (argv[i][0] == '-' && argv[i][1]) continue;
      if (std::strchr(argv[i], '=')) continue;
      v = argv[i];
      reg(std::to_string(nfiles++), &v);
    }
  }

  // rc_get_file (readconfig.c:1056-1082)
  static int get_file(int argc, const char* const* argv) {
    bool ignore_hyphen = false;
    for (int i = 1; i < argc; ++i) {
      const char* c = argv[i];
      if (!ignore_hyphen && c[0] == '-') {
        if (std::strcmp(c, "--")) ignore_hyphen = true;    // sic
        continue;
      }
      if (!std::strchr(c, '=') && std::strstr(c, ".cfg")) return i;
    }
    return 0;
  }

  // rc_register_args (readconfig.c:900-953)
  void register_args(int argc, const char* const* argv) {
    for (int i = 1; i < argc; ++i) {
      if (argv[i][0] == '-' && argv[i][1]) {
        reg(argv[i] + 1, nullptr);
        continue;
      }
      for (const char* c = argv[i]; *c; ++c) {
        if (*c != '=') continue;
        std::string param(argv[i], c), value;
        if (c[1] == '$') {
          const Entry* s = find_plain(c + 2);
          if (!s) break;      // the reference never returns from this case; the argument is dropped here
          value = s->has_value ? s->value : std::string("1");
        } else {
          value = c + 1;
        }
        reg(param, &value);
      }
    }
  }

  // rc_sprint (readconfig.c:1113-1253), the non-"classic" format the reference is built with
  std::string sprint() const {
    std::string out;
    for (const Entry& e : entries) {
      if (!out.empty()) out += "; ";
      out += e.param;
      if (!e.has_value) continue;
      if (e.m > 0 || e.n > 0) out += "[" + std::to_string(e.m) + "][" + std::to_string(e.n) + "]";
      out.push_back('=');
      bool wrap = false;
      for (unsigned char ch : e.value) if (ch <= ' ') wrap = true;
      if (wrap) out += "{" + compress_whitespace(e.value) + "}";
      else out += e.value;
    }
    return out;
  }
};

namespace {

int copy_out(const std::string& s, char* buf, size_t cap, size_t* len) {
  if (len) *len = s.size();
  if (buf && cap > 0) {
    const size_t n = s.size() < cap - 1 ? s.size() : cap - 1;
    std::memcpy(buf, s.data(), n);
    buf[n] = '\0';
  }
  return ECCKD_OK;
}

}  // namespace

extern "C" {

int ecckd_cfg_create(ecckd_cfg** out) {
  ECCKD_REQUIRE(out, "ecckd_cfg_create: NULL argument");
  *out = new ecckd_cfg();
  return ECCKD_OK;
}

int ecckd_cfg_from_args(int argc, const char* const* argv, ecckd_cfg** out) {
  ECCKD_REQUIRE(out && argc >= 1 && argv, "ecckd_cfg_from_args: bad argument");
  ecckd_cfg* c = new ecckd_cfg();
  c->register_files(argc, argv);
  const int ifile = ecckd_cfg::get_file(argc, argv);
  if (ifile) {
    const int rc = c->append_file(argv[ifile]);
    if (rc != ECCKD_OK) { delete c; return ECCKD_CANNOT_OPEN_MANDATORY_FILE; }
    c->file_name = argv[ifile];
    if (c->have_section) {
      const int rc2 = ecckd::fail(ECCKD_CANNOT_OPEN_MANDATORY_FILE, "Section \"%s\" unterminated by \\end", c->section.c_str());
      delete c;
      return rc2;
    }
  }
  c->register_args(argc, argv);
  *out = c;
  return ECCKD_OK;
}

int ecckd_cfg_append_file(ecckd_cfg* c, const char* path) {
  ECCKD_REQUIRE(c && path, "ecckd_cfg_append_file: NULL argument");
  if (c->append_file(path) != ECCKD_OK) return ECCKD_CANNOT_OPEN_MANDATORY_FILE;
  if (c->have_section) {   // rc_read (readconfig.c:1035-1043)
    std::string s = c->section;
    c->section.clear(); c->have_section = false;
    return ecckd::fail(ECCKD_CANNOT_OPEN_MANDATORY_FILE, "Section \"%s\" unterminated by \\end", s.c_str());
  }
  return ECCKD_OK;
}

int ecckd_cfg_append_text(ecckd_cfg* c, const char* text, const char* name) {
  ECCKD_REQUIRE(c && text, "ecckd_cfg_append_text: NULL argument");
  return c->append_text(text, name ? name : "<text>");
}

int ecckd_cfg_register(ecckd_cfg* c, const char* param, const char* value) {
  ECCKD_REQUIRE(c && param, "ecckd_cfg_register: NULL argument");
  if (value) { std::string v = value; c->reg(param, &v); }
  else c->reg(param, nullptr);
  return ECCKD_OK;
}

int ecckd_cfg_destroy(ecckd_cfg* c) {
  delete c;
  return ECCKD_OK;
}

int ecckd_cfg_file_name(const ecckd_cfg* c, char* buf, size_t cap, size_t* len) {
  ECCKD_REQUIRE(c, "ecckd_cfg_file_name: NULL argument");
  return copy_out(c->file_name, buf, cap, len);
}

int ecckd_cfg_count(const ecckd_cfg* c, int* n) {
  ECCKD_REQUIRE(c && n, "ecckd_cfg_count: NULL argument");
  *n = (int)c->entries.size();
  return ECCKD_OK;
}

int ecckd_cfg_entry(const ecckd_cfg* c, int i, char* param, size_t param_cap, char* value, size_t value_cap, size_t* value_len,
                    int* has_value, int* m, int* n) {
  ECCKD_REQUIRE(c && i >= 0 && (size_t)i < c->entries.size(), "ecckd_cfg_entry: index %d outside the configuration", i);
  const Entry& e = c->entries[i];
  copy_out(e.param, param, param_cap, nullptr);
  copy_out(e.value, value, value_cap, value_len);
  if (has_value) *has_value = e.has_value ? 1 : 0;
  if (m) *m = e.m;
  if (n) *n = e.n;
  return ECCKD_OK;
}

int ecckd_cfg_exists(const ecckd_cfg* c, const char* scope, const char* param, int* exists) {
  ECCKD_REQUIRE(c && param && exists, "ecckd_cfg_exists: NULL argument");
  *exists = c->find(scope, param) != nullptr;
  return ECCKD_OK;
}

// rc_get_boolean (readconfig.c:1262-1287)
int ecckd_cfg_get_boolean(const ecckd_cfg* c, const char* scope, const char* param, int* value) {
  ECCKD_REQUIRE(c && param && value, "ecckd_cfg_get_boolean: NULL argument");
  const Entry* e = c->find(scope, param);
  if (!e) { *value = 0; return ECCKD_OK; }
  if (!e->has_value) { *value = 1; return ECCKD_OK; }
  const char* v = e->value.c_str();
  if (strncasecmp(v, "false", 5) == 0 || strncasecmp(v, "no", 2) == 0) { *value = 0; return ECCKD_OK; }
  char* end = nullptr;
  const double x = std::strtod(v, &end);
  *value = (end == v || x != 0.0) ? 1 : 0;
  return ECCKD_OK;
}

// rc_assign_int (readconfig.c:1293-1349): *value untouched when not found / not a number
int ecckd_cfg_get_int(const ecckd_cfg* c, const char* scope, const char* param, int* value, int* found) {
  ECCKD_REQUIRE(c && param && value && found, "ecckd_cfg_get_int: NULL argument");
  *found = 0;
  const Entry* e = c->find(scope, param);
  if (!e || !e->has_value) return ECCKD_OK;
  char* end = nullptr;
  const long v = std::strtol(e->value.c_str(), &end, 10);
  if (end == e->value.c_str()) return ECCKD_OK;
  *value = (int)v; *found = 1;
  return ECCKD_OK;
}

int ecckd_cfg_get_real(const ecckd_cfg* c, const char* scope, const char* param, double* value, int* found) {
  ECCKD_REQUIRE(c && param && value && found, "ecckd_cfg_get_real: NULL argument");
  *found = 0;
  const Entry* e = c->find(scope, param);
  if (!e || !e->has_value) return ECCKD_OK;
  char* end = nullptr;
  const double v = std::strtod(e->value.c_str(), &end);
  if (end == e->value.c_str()) return ECCKD_OK;
  *value = v; *found = 1;
  return ECCKD_OK;
}

// rc_get_string (isub < 0, trailing whitespace removed) / rc_get_substring (isub >= 0)
int ecckd_cfg_get_string(const ecckd_cfg* c, const char* scope, const char* param, int isub, char* buf, size_t cap, size_t* len,
                         int* found) {
  ECCKD_REQUIRE(c && param && found, "ecckd_cfg_get_string: NULL argument");
  *found = 0;
  if (len) *len = 0;
  const Entry* e = c->find(scope, param);
  if (!e || !e->has_value) return ECCKD_OK;
  std::string s;
  if (isub < 0) { s = e->value; strip_trailing(s); }
  else if (!substring(e->value, isub, s)) return ECCKD_OK;
  *found = 1;
  return copy_out(s, buf, cap, len);
}

// rc_size (readconfig.c:1653-1666)
int ecckd_cfg_size(const ecckd_cfg* c, const char* scope, const char* param, int* count, int* m, int* n) {
  ECCKD_REQUIRE(c && param && count, "ecckd_cfg_size: NULL argument");
  const Entry* e = c->find(scope, param);
  *count = 0;
  if (m) *m = 0;
  if (n) *n = 0;
  if (!e || !e->has_value) return ECCKD_OK;
  *count = count_substrings(e->value);
  if (m) *m = e->m;
  if (n) *n = e->n;
  return ECCKD_OK;
}

// rc_get_real_vector (readconfig.c:1758-1790): numbers are taken until the first item strtod rejects
int ecckd_cfg_get_real_vector(const ecckd_cfg* c, const char* scope, const char* param, double* buf, int cap, int* len) {
  ECCKD_REQUIRE(c && param && len, "ecckd_cfg_get_real_vector: NULL argument");
  *len = 0;
  const Entry* e = c->find(scope, param);
  if (!e || !e->has_value) return ECCKD_OK;
  const char* p = e->value.c_str();
  while (*p) {
    char* end = nullptr;
    const double v = std::strtod(p, &end);
    if (end == p) break;
    if (buf && *len < cap) buf[*len] = v;
    ++*len;
    p = end;
  }
  return ECCKD_OK;
}

int ecckd_cfg_get_int_vector(const ecckd_cfg* c, const char* scope, const char* param, int* buf, int cap, int* len) {
  ECCKD_REQUIRE(c && param && len, "ecckd_cfg_get_int_vector: NULL argument");
  *len = 0;
  const Entry* e = c->find(scope, param);
  if (!e || !e->has_value) return ECCKD_OK;
  const char* p = e->value.c_str();
  while (*p) {
    char* end = nullptr;
    const long v = std::strtol(p, &end, 10);
    if (end == p) break;
    if (buf && *len < cap) buf[*len] = (int)v;
    ++*len;
    p = end;
  }
  return ECCKD_OK;
}

int ecckd_cfg_sprint(const ecckd_cfg* c, char* buf, size_t cap, size_t* len) {
  ECCKD_REQUIRE(c, "ecckd_cfg_sprint: NULL argument");
  return copy_out(c->sprint(), buf, cap, len);
}

}  // extern "C"
