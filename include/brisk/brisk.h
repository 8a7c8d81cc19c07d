// brisk/brisk.h - umbrella header of the MI355X BRISK engine's host classes
// (same role as the reference's brisk/include/brisk/brisk.h:44-65, restricted to the AGAST detect + describe path).
#ifndef BRISK_BRISK_H_
#define BRISK_BRISK_H_

#include <agast/wrap-opencv.h>
#include <brisk/brisk-descriptor-extractor.h>
#include <brisk/brisk-feature-detector.h>
#include <brisk/brute-force-matcher.h>

namespace cv {
typedef brisk::BriskDescriptorExtractor BriskDescriptorExtractor;
typedef brisk::BriskFeatureDetector BriskFeatureDetector;
}  // namespace cv

#endif  // BRISK_BRISK_H_
