// test scaffolding -- see README.md: the member names of AllMutations / HapChrom / HapGenome / HapSet the shims read
#pragma once
#include <deque>
#include <string>
#include <vector>
#include "ref_classes.h"
struct AllMutations {
    std::deque<uint64> old_pos, new_pos;
    std::deque<char*> nucleos;            // nullptr = deletion
    size_t size() const { return old_pos.size(); }
};
struct HapChrom {
    AllMutations mutations;
    uint64 chrom_size = 0;
};
struct HapGenome {
    std::string name;
    std::vector<HapChrom> chromosomes;
    const HapChrom& operator[](const uint64& i) const { return chromosomes[i]; }
    uint64 size() const { return chromosomes.size(); }
};
struct HapSet {
    std::vector<HapGenome> haplotypes;
    const RefGenome* reference = nullptr;
    const HapGenome& operator[](const uint64& i) const { return haplotypes[i]; }
    uint64 size() const { return haplotypes.size(); }
};
