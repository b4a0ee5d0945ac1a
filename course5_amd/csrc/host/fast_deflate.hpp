#pragma once
// deflate for doubles that were widened from floats (the .vti image): see fast_deflate.cpp
#include <cstddef>
#include <cstdint>

namespace c5 {
// Writes `count` doubles as one zlib stream into out[0, cap).  Returns the stream's size, or 0 when the input is not
// made of widened floats (some value has a bit set among its low 24), has fewer than two values, or does not fit cap:
// the caller then uses the general compressor.
size_t deflate_widened_doubles(const double* vals, size_t count, unsigned char* out, size_t cap);
// The same stream, straight from the floats (what static_cast<double> makes of them), without the array of doubles.
size_t deflate_floats_as_doubles(const float* vals, size_t count, unsigned char* out, size_t cap);
}  // namespace c5
