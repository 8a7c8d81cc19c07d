// The drop-in classes on images whose candidate / tie / keypoint counts exceed the workspace's default capacities:
// the classes grow the workspace and repeat the call (the reference has no capacities), the result is written to a
// file and compared with the oracle by tests/test_cpp_classes.py.
// usage: test_capacity <raw u8 image> <cols> <rows> <threshold> <octaves> <out file>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include <brisk/brisk.h>

int main(int argc, char** argv) {
  if (argc < 7) return 3;
  const int cols = atoi(argv[2]), rows = atoi(argv[3]), thr = atoi(argv[4]), octaves = atoi(argv[5]);
  agast::Mat img(rows, cols, CV_8UC1);
  FILE* f = fopen(argv[1], "rb");
  if (!f || fread(img.data, 1, (size_t)rows * cols, f) != (size_t)rows * cols) return 3;
  fclose(f);
  try {
    brisk::BriskFeatureDetector det(thr, octaves);
    brisk::BriskDescriptorExtractor ext;
    std::vector<agast::KeyPoint> kps;
    det.detect(img, kps);
    const size_t ndet = kps.size();
    std::vector<agast::KeyPoint> detected = kps;
    agast::Mat desc;
    ext.compute(img, kps, desc);
    FILE* o = fopen(argv[6], "wb");
    if (!o) return 3;
    const int hdr[4] = {(int)ndet, (int)kps.size(), desc.rows, desc.cols};
    fwrite(hdr, sizeof(int), 4, o);
    fwrite(detected.data(), sizeof(agast::KeyPoint), ndet, o);
    fwrite(kps.data(), sizeof(agast::KeyPoint), kps.size(), o);
    for (int r = 0; r < desc.rows; ++r) fwrite(desc.data + (size_t)r * desc.step, 1, (size_t)desc.cols, o);
    fclose(o);
    printf("detected %zu described %zu\n", ndet, kps.size());
  } catch (const std::exception& e) {
    printf("exception: %s\n", e.what());
    return 2;
  }
  return 0;
}
