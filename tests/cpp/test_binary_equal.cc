// test_binary_equal.cc - the reference's golden test (brisk/src/test/test-binary-equal.cc:319-333,
// bench-ds.h:311-430) written against the drop-in host classes of the MI355X engine.
// Reads brisk_verification_ast.set, runs BriskFeatureDetector(70) + BriskDescriptorExtractor() on each stored
// image and requires every keypoint field to be exactly equal and every descriptor row identical (the reference
// tolerates a Hamming distance of 5; here 0 is required).  Also checks the Harris set's descriptors on the
// externally provided keypoints.  Exit code 0 = verification success.
#include <brisk/brisk.h>

#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

struct Entry {
  std::string path;
  agast::Mat image;
  std::vector<agast::KeyPoint> keypoints;
  agast::Mat descriptors;
};

template <typename T> static T rd(std::ifstream& in) {
  T v;
  in.read(reinterpret_cast<char*>(&v), sizeof(T));
  return v;
}

static agast::Mat read_mat(std::ifstream& in) {
  const int rows = rd<int>(in), cols = rd<int>(in), type = rd<int>(in), esz = rd<int>(in);
  agast::Mat m(rows, cols * esz, CV_8UC1);
  (void)type;
  in.read(reinterpret_cast<char*>(m.data), (std::streamsize)rows * cols * esz);
  return m;
}

// .set layout: brisk/src/test/serialization.cc:46-149, bench-ds.cc:57-94
static std::vector<Entry> read_set(const std::string& fn) {
  std::ifstream in(fn.c_str(), std::ios::binary);
  if (!in.good()) throw std::runtime_error("cannot open " + fn);
  std::vector<Entry> out(rd<uint32_t>(in));
  for (Entry& e : out) {
    const uint32_t len = rd<uint32_t>(in);
    e.path.resize(len);
    in.read(&e.path[0], len);
    e.image = read_mat(in);
    e.keypoints.resize(rd<uint32_t>(in));
    for (agast::KeyPoint& k : e.keypoints) {
      k.angle = rd<float>(in);
      k.class_id = rd<int>(in);
      k.octave = rd<int>(in);
      k.pt.x = rd<float>(in);
      k.pt.y = rd<float>(in);
      k.response = rd<float>(in);
      k.size = rd<float>(in);
    }
    e.descriptors = read_mat(in);
    const uint32_t nblobs = rd<uint32_t>(in);
    for (uint32_t b = 0; b < nblobs; ++b) {
      const uint32_t kl = rd<uint32_t>(in);
      in.seekg(kl, std::ios::cur);
      const uint32_t sz = rd<uint32_t>(in);
      in.seekg(sz, std::ios::cur);
    }
  }
  return out;
}

static bool same_kp(const agast::KeyPoint& a, const agast::KeyPoint& b) { return memcmp(&a, &b, sizeof(a)) == 0; }

int main(int argc, char** argv) {
  const std::string dir = argc > 1 ? argv[1] : "tests/golden";
  int failures = 0;
  try {
    {  // TEST(Brisk, ValidationAST)
      std::vector<Entry> ds = read_set(dir + "/brisk_verification_ast.set");
      brisk::BriskFeatureDetector detector(70);
      brisk::BriskDescriptorExtractor extractor;
      for (Entry& e : ds) {
        std::vector<agast::KeyPoint> kps;
        agast::Mat desc;
        detector.detect(e.image, kps);
        extractor.compute(e.image, kps, desc);
        bool ok = kps.size() == e.keypoints.size() && desc.rows == e.descriptors.rows && desc.cols == e.descriptors.cols;
        for (size_t i = 0; ok && i < kps.size(); ++i) ok = same_kp(kps[i], e.keypoints[i]);
        for (int r = 0; ok && r < desc.rows; ++r)
          ok = memcmp(desc.data + (size_t)r * desc.step, e.descriptors.data + (size_t)r * e.descriptors.step, desc.cols) == 0;
        std::printf("AST    %-24s keypoints %zu / %zu  %s\n", e.path.c_str(), kps.size(), e.keypoints.size(), ok ? "OK" : "MISMATCH");
        failures += ok ? 0 : 1;
      }
    }
    {  // descriptors of the Harris golden set on its stored keypoints (orientation re-estimated)
      std::vector<Entry> ds = read_set(dir + "/brisk_verification_harris.set");
      brisk::BriskDescriptorExtractor extractor(true, true);
      for (Entry& e : ds) {
        std::vector<agast::KeyPoint> kps = e.keypoints;
        for (agast::KeyPoint& k : kps) k.angle = -1;
        std::vector<std::bitset<384> > bits;
        extractor.compute(e.image, kps, bits);
        bool ok = kps.size() == e.keypoints.size();
        for (size_t i = 0; ok && i < kps.size(); ++i) {
          ok = same_kp(kps[i], e.keypoints[i]);
          for (unsigned b = 0; ok && b < 384; ++b)
            ok = bits[i][b] == ((e.descriptors.data[i * e.descriptors.step + (b >> 3)] >> (b & 7)) & 1);
        }
        std::printf("HARRIS %-24s keypoints %zu / %zu  %s\n", e.path.c_str(), kps.size(), e.keypoints.size(), ok ? "OK" : "MISMATCH");
        failures += ok ? 0 : 1;
      }
    }
  } catch (const std::exception& ex) {
    std::printf("exception: %s\n", ex.what());
    return 2;
  }
  std::printf(failures ? "******* Verification failed *******\n" : "******* Verification success *******\n");
  return failures ? 1 : 0;
}
