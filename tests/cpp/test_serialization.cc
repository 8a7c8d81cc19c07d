// test_serialization.cc - the `.set` container in C++ (tests/cpp/set_serialization.h), mirroring the reference's
// test-serialization.cc: element round trips, and the reference's own golden files read and written back
// byte-identically.  Needs no GPU.  Usage: test_serialization <golden dir> <scratch file>.
#include <cstdio>
#include <cstring>
#include <iterator>
#include <sstream>

#include "set_serialization.h"

static std::string slurp(const std::string& fn) {
  std::ifstream in(fn.c_str(), std::ios::binary);
  return std::string(std::istreambuf_iterator<char>(in), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv) {
  const std::string dir = argc > 1 ? argv[1] : "tests/golden";
  const std::string tmp = argc > 2 ? argv[2] : "/tmp/brisk_set_roundtrip.set";
  int failures = 0;
  try {
    {  // scalars, strings, key points through a memory stream
      std::stringstream ss;
      agast::KeyPoint k;
      k.pt.x = 12.5f; k.pt.y = -3.25f; k.size = 17.f; k.angle = 271.5f; k.response = 99.f; k.octave = 3; k.class_id = 7;
      setio::Serialize(std::string("a path/with spaces.pgm"), &ss);
      setio::Serialize(k, &ss);
      setio::Put<uint32_t>(0xDEADBEEFu, &ss);
      std::string s;
      agast::KeyPoint k2;
      uint32_t u = 0;
      setio::DeSerialize(&s, &ss);
      setio::DeSerialize(&k2, &ss);
      setio::Get(&u, &ss);
      const bool ok = s == "a path/with spaces.pgm" && memcmp(&k, &k2, sizeof(k)) == 0 && u == 0xDEADBEEFu;
      std::printf("elements            %s\n", ok ? "OK" : "MISMATCH");
      failures += !ok;
    }
    for (const char* name : {"brisk_verification_ast.set", "brisk_verification_harris.set"}) {
      const std::string fn = dir + "/" + name;
      const std::vector<setio::DatasetEntry> set = setio::ReadSet(fn);
      setio::WriteSet(tmp, set);
      const std::string a = slurp(fn), b = slurp(tmp);
      size_t nk = 0;
      for (const setio::DatasetEntry& e : set) nk += e.keypoints.size();
      const bool ok = !a.empty() && a == b && set.size() == 2 && nk > 800;
      std::printf("%-32s %zu entries, %zu keypoints, %zu bytes  %s\n", name, set.size(), nk, a.size(), ok ? "OK" : "MISMATCH");
      failures += !ok;
    }
    std::remove(tmp.c_str());
  } catch (const std::exception& e) {
    std::printf("%s\n", e.what());
    return 2;
  }
  return failures ? 1 : 0;
}
