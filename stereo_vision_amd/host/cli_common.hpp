// cli_common.hpp -- argv handling shared by the CLI programs: the reference's positional
// arguments first, then `--name value` options for what the reference hard-codes in its mains.
#ifndef SGMX_CLI_COMMON_HPP
#define SGMX_CLI_COMMON_HPP

#include <cstdlib>
#include <iostream>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace sgmx {
namespace cli {

struct Args {
    std::vector<std::string> pos;
    std::map<std::string, std::string> opt;
    bool has(const std::string &k) const { return opt.count(k) != 0; }
    std::string str(const std::string &k, const std::string &def) const
    {
        auto it = opt.find(k);
        return it == opt.end() ? def : it->second;
    }
    int num(const std::string &k, int def) const
    {
        auto it = opt.find(k);
        if (it == opt.end())
            return def;
        size_t used = 0;
        int v = 0;
        try {
            v = std::stoi(it->second, &used);
        } catch (const std::exception &) {
            used = 0;
        }
        if (used != it->second.size() || it->second.empty())
            throw std::invalid_argument("option --" + k + " wants an integer, got '" + it->second + "'");
        return v;
    }
};

// `flags` are options without a value; `short_map` maps one-letter options to long names.
inline Args parse(int argc, char **argv, const std::vector<std::string> &flags = {},
                  const std::map<char, std::string> &short_map = {})
{
    Args a;
    for (int i = 1; i < argc; i++) {
        std::string s = argv[i];
        std::string key;
        if (s.size() > 2 && s[0] == '-' && s[1] == '-')
            key = s.substr(2);
        else if (s.size() == 2 && s[0] == '-' && short_map.count(s[1]))
            key = short_map.at(s[1]);
        if (key.empty()) {
            a.pos.push_back(s);
            continue;
        }
        const size_t eq = key.find('=');
        if (eq != std::string::npos) {
            a.opt[key.substr(0, eq)] = key.substr(eq + 1);
            continue;
        }
        bool is_flag = false;
        for (auto &f : flags)
            is_flag = is_flag || f == key;
        if (is_flag) {
            a.opt[key] = "1";
            continue;
        }
        if (i + 1 >= argc)
            throw std::invalid_argument("option --" + key + " wants a value");
        a.opt[key] = argv[++i];
    }
    return a;
}

inline int to_int(const std::string &s, const char *what)
{
    size_t used = 0;
    int v = 0;
    try {
        v = std::stoi(s, &used);
    } catch (const std::exception &) {
        used = 0;
    }
    if (s.empty() || used != s.size())
        throw std::invalid_argument(std::string(what) + " wants an integer, got '" + s + "'");
    return v;
}

// Runs body(), turning exceptions into a message on stderr and exit status 2 (usage errors are 1,
// as in the reference's mains).
template <class F> inline int guarded(const char *prog, F body)
{
    try {
        return body();
    } catch (const std::exception &e) {
        std::cerr << prog << ": " << e.what() << std::endl;
        return 2;
    }
}

} // namespace cli
} // namespace sgmx
#endif
