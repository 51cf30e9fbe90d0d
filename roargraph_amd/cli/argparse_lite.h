// argparse_lite.h -- tiny argv parser with the option spellings of the reference's boost::program_options CLIs
// (--name value, --name=value, short aliases, multitoken options).  Boost is not available in this image.
#pragma once
#include <cstdlib>
#include <iostream>
#include <map>
#include <set>
#include <string>
#include <vector>

class Args {
   public:
    void add(const std::string &name, bool required, const std::string &help, const std::string &def = "",
             const std::string &alias = "") {
        spec_[name] = {required, help, def};
        if (!alias.empty()) alias_[alias] = name;
        order_.push_back(name);
    }
    // returns false (after printing what boost would: "the option '--x' is required but missing") on error
    bool parse(int argc, char **argv) {
        std::string cur;
        for (int i = 1; i < argc; ++i) {
            std::string a = argv[i];
            const bool negnum = a.size() > 1 && a[0] == '-' && (isdigit((unsigned char)a[1]) || a[1] == '.');
            if (a.size() > 1 && a[0] == '-' && !negnum) {
                std::string name = a[1] == '-' ? a.substr(2) : a.substr(1), val;
                const size_t eq = name.find('=');
                bool has_val = false;
                if (eq != std::string::npos) { val = name.substr(eq + 1); name = name.substr(0, eq); has_val = true; }
                if (alias_.count(name)) name = alias_[name];
                if (name == "help" || name == "h") { help_ = true; return true; }
                if (!spec_.count(name)) { std::cerr << "unrecognised option '" << a << "'\n"; return false; }
                cur = name;
                seen_.insert(name);
                if (has_val) vals_[name].push_back(val);
            } else {
                if (cur.empty()) { std::cerr << "too many positional options have been specified on the command line\n"; return false; }
                vals_[cur].push_back(a);
            }
        }
        for (auto &kv : spec_)
            if (kv.second.required && !vals_.count(kv.first) && kv.second.def.empty()) {
                std::cerr << "the option '--" << kv.first << "' is required but missing\n";
                return false;
            }
        return true;
    }
    bool help() const { return help_; }
    void usage(std::ostream &os) const {
        os << "Arguments:\n";
        for (auto &n : order_) os << "  --" << n << "  " << spec_.at(n).help << "\n";
    }
    bool has(const std::string &n) const { return vals_.count(n) > 0; }
    std::string str(const std::string &n) const {
        auto it = vals_.find(n);
        if (it != vals_.end() && !it->second.empty()) return it->second.back();
        return spec_.at(n).def;
    }
    std::vector<std::string> list(const std::string &n) const {
        auto it = vals_.find(n);
        return it == vals_.end() ? std::vector<std::string>() : it->second;
    }
    unsigned long u(const std::string &n) const { return std::strtoul(str(n).c_str(), nullptr, 10); }

   private:
    struct Spec { bool required; std::string help, def; };
    std::map<std::string, Spec> spec_;
    std::map<std::string, std::string> alias_;
    std::map<std::string, std::vector<std::string>> vals_;
    std::set<std::string> seen_;
    std::vector<std::string> order_;
    bool help_ = false;
};
