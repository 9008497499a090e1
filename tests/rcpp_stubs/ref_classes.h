// test scaffolding -- see README.md: the member names of RefChrom / RefGenome the shims read
#pragma once
#include <deque>
#include <string>
#include "jackalope_types.h"
struct RefChrom {
    std::string name, nucleos;
    uint64 size() const { return nucleos.size(); }
};
struct RefGenome {
    std::deque<RefChrom> chromosomes;
    std::string name = "REF";
    const RefChrom& operator[](const uint64& i) const { return chromosomes[i]; }
    uint64 size() const { return chromosomes.size(); }
};
