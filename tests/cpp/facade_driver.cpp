// facade_driver.cpp — a caller written against the REFERENCE's interface (the shape of src/test/main.cpp:13-35:
// line 1 of stdin = text, line 2 = pattern; prints "is match? 0/1"), compiled against include/rregex.hpp.
// Extra lines of stdin are further texts; each gets its own iterator, like repeated get_acceptance_iter calls.
#include <cstdio>
#include <iostream>
#include <string>
#include <vector>

#include "rregex.hpp"

using namespace Regex;

int main() {
    std::string text, pattern;
    if (!std::getline(std::cin, text) || !std::getline(std::cin, pattern)) return 2;
    try {
        RRegex r(pattern.c_str());
        std::vector<std::string> texts{text};
        for (std::string more; std::getline(std::cin, more);) texts.push_back(more);
        for (auto &t : texts) {
            std::vector<char> buf(t.begin(), t.end());
            buf.push_back('\0');
            auto before = r.get_acceptance_iter(buf.data());
            bool nullable = (*before).has_value();
            auto acceptance_iter = r.get_acceptance_iter(buf.data())++;
            bool is_match = (*acceptance_iter).has_value();
            IteratorWrapper copy(acceptance_iter);               // deep copy through create_copy()
            copy++;                                              // idempotent after the terminator
            bool again = (*copy).has_value();
            std::cout << "is match? " << is_match << " nullable " << nullable << " again " << again;
            if (is_match) std::cout << " len " << (*acceptance_iter)->str().size();
            std::cout << std::endl;
        }
        // batch entry on the same pattern
        std::string blob;
        for (auto &t : texts) { blob += t; blob += '\n'; }
        auto acc = r.match_lines(blob.data(), blob.size());
        std::cout << "batch";
        for (auto a : acc) std::cout << ' ' << int(a);
        std::cout << std::endl;
    } catch (const std::runtime_error &e) {
        std::cout << "error: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
