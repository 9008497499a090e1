// test scaffolding -- see README.md
#pragma once
#include <cstdlib>
#include <string>
inline void expand_path(std::string& f) {          // R's path.expand: a leading "~"
    if (!f.empty() && f[0] == '~') { const char* h = std::getenv("HOME"); if (h) f = std::string(h) + f.substr(1); }
}
