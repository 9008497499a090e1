// test scaffolding -- see README.md
#pragma once
#include <cstdint>
typedef uint_fast8_t uint8;
typedef uint_fast32_t uint32;
typedef int_fast64_t sint64;
typedef uint_fast64_t uint64;
