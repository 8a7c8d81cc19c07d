// brisk_pattern.h - host-side construction of the descriptor pattern tables
// (brisk/src/brisk-descriptor-extractor.cc:65-343).  Host libm is used on purpose: the reference
// builds these tables on the host, so building them the same way makes them bit-identical.
#pragma once
#include <string>
#include <vector>

#include "brisk_common.h"

struct BriskPatternHost {
  int npoints = 0, nshort = 0, nlong = 0, strings = 0;
  int basicscale = 0;               // scale index used when scaleInvariance == false (:631-635)
  std::vector<float> scale_list;    // [64]
  std::vector<int> size_list;       // [64]
  std::vector<float> size_thresh;   // [64]
  std::vector<float> mult;          // [64][npoints]
  std::vector<float> sigma;         // [64][npoints]
  std::vector<int> scaling;         // [64][npoints][2]
  std::vector<double> uv;           // [1024][npoints][2]
  std::vector<uint16_t> short_pairs;  // [nshort][2]
  std::vector<int> long_pairs;        // [nlong][4]
};

// version 2 with the built-in table, or version 1 (generated). Returns false on bad version.
bool brisk_pattern_build_default(int version, float pattern_scale, BriskPatternHost* out, std::string* err);
// version-2 style pattern from .ptn text
bool brisk_pattern_build_from_text(const char* text, float pattern_scale, BriskPatternHost* out, std::string* err);
// scale index by the reference expression (:636-646), used to derive size_thresh and by tests
int brisk_pattern_scale_index_host(float size);
