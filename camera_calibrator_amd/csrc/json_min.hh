// json_min.hh -- the small subset of JSON the ExtrinsicsCalibrator wire format needs (objects,
// arrays, numbers, booleans, null). The reference uses nlohmann::json (conanfile.txt:14), which is
// not part of this build. Output matches nlohmann's compact dump: keys in alphabetical order, no
// whitespace, NaN -> null.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace jsonmin {

struct Value {
  enum Kind { Null, Bool, Number, Array, Object } kind = Null;
  bool b = false;
  bool is_int = false;  // written without a fractional part (ids)
  double num = 0.0;
  std::vector<Value> arr;
  std::map<std::string, Value> obj;  // std::map: keys sorted like nlohmann's default object type

  static Value number(double v) { Value x; x.kind = Number; x.num = v; return x; }
  static Value integer(unsigned long long v) { Value x; x.kind = Number; x.num = (double)v; x.is_int = true; return x; }
  static Value boolean(bool v) { Value x; x.kind = Bool; x.b = v; return x; }
  static Value array() { Value x; x.kind = Array; return x; }
  static Value object() { Value x; x.kind = Object; return x; }

  const Value& at(const std::string& key) const {
    auto it = obj.find(key);
    if (kind != Object || it == obj.end()) throw std::runtime_error("json: missing key '" + key + "'");
    return it->second;
  }
  const Value& at(size_t i) const {
    if (kind != Array || i >= arr.size()) throw std::runtime_error("json: index out of range");
    return arr[i];
  }
  size_t size() const { return kind == Array ? arr.size() : obj.size(); }
  double as_number() const {
    if (kind != Number) throw std::runtime_error("json: number expected");
    return num;
  }
  bool as_bool() const {
    if (kind != Bool) throw std::runtime_error("json: boolean expected");
    return b;
  }
};

inline void dump(const Value& v, std::string& out) {
  switch (v.kind) {
    case Value::Null: out += "null"; break;
    case Value::Bool: out += v.b ? "true" : "false"; break;
    case Value::Number: {
      if (!std::isfinite(v.num)) { out += "null"; break; }  // nlohmann writes non-finite numbers as null
      if (v.is_int) { out += std::to_string((unsigned long long)v.num); break; }
      if (v.num == std::floor(v.num) && std::fabs(v.num) < 9.0e15 && !(v.num == 0.0 && std::signbit(v.num))) {
        char buf[32];
        std::snprintf(buf, sizeof(buf), "%.1f", v.num);  // integral floats print as "1.0"
        out += buf;
      } else {
        char buf[40];
        for (int prec = 15; prec <= 17; ++prec) {  // shortest representation that round-trips
          std::snprintf(buf, sizeof(buf), "%.*g", prec, v.num);
          if (std::strtod(buf, nullptr) == v.num) break;
        }
        out += buf;
      }
      break;
    }
    case Value::Array: {
      out += '[';
      for (size_t i = 0; i < v.arr.size(); ++i) { if (i) out += ','; dump(v.arr[i], out); }
      out += ']';
      break;
    }
    case Value::Object: {
      out += '{';
      bool first = true;
      for (const auto& kv : v.obj) {
        if (!first) out += ',';
        first = false;
        out += '"'; out += kv.first; out += "\":";
        dump(kv.second, out);
      }
      out += '}';
      break;
    }
  }
}

class Parser {
 public:
  explicit Parser(const std::string& text) : s_(text) {}
  Value parse() {
    Value v = value();
    ws();
    if (i_ != s_.size()) fail("trailing characters");
    return v;
  }

 private:
  [[noreturn]] void fail(const char* what) const { throw std::runtime_error(std::string("json parse error: ") + what + " at offset " + std::to_string(i_)); }
  void ws() { while (i_ < s_.size() && (s_[i_] == ' ' || s_[i_] == '\n' || s_[i_] == '\t' || s_[i_] == '\r')) ++i_; }
  bool eat(char c) { ws(); if (i_ < s_.size() && s_[i_] == c) { ++i_; return true; } return false; }
  bool word(const char* w) { size_t n = std::char_traits<char>::length(w); if (s_.compare(i_, n, w) == 0) { i_ += n; return true; } return false; }
  Value value() {
    ws();
    if (i_ >= s_.size()) fail("unexpected end");
    const char c = s_[i_];
    if (c == '{') {
      ++i_;
      Value v = Value::object();
      if (eat('}')) return v;
      do {
        ws();
        if (i_ >= s_.size() || s_[i_] != '"') fail("key expected");
        const size_t e = s_.find('"', i_ + 1);
        if (e == std::string::npos) fail("unterminated key");
        const std::string key = s_.substr(i_ + 1, e - i_ - 1);
        i_ = e + 1;
        if (!eat(':')) fail("':' expected");
        v.obj[key] = value();
      } while (eat(','));
      if (!eat('}')) fail("'}' expected");
      return v;
    }
    if (c == '[') {
      ++i_;
      Value v = Value::array();
      if (eat(']')) return v;
      do { v.arr.push_back(value()); } while (eat(','));
      if (!eat(']')) fail("']' expected");
      return v;
    }
    if (word("null")) return Value();
    if (word("true")) return Value::boolean(true);
    if (word("false")) return Value::boolean(false);
    char* end = nullptr;
    const double d = std::strtod(s_.c_str() + i_, &end);
    if (end == s_.c_str() + i_) fail("value expected");
    i_ = (size_t)(end - s_.c_str());
    return Value::number(d);
  }
  const std::string& s_;
  size_t i_ = 0;
};

}  // namespace jsonmin
