// brisk/internal/image-functions-16.h - the reference's 16-bit image functions over the MI355X engine's C ABI:
//   brisk::Halfsample16 / brisk::Twothirdsample16   brisk/include/brisk/internal/image-down-sampling.h, src/image-down-sampling.cc:56-139, 394-548
//   brisk::IntegralImage16                          brisk/include/brisk/internal/integral-image.h:163-218
// Same names, arguments and size requirements as the reference (the destination of the down-samplers must have the
// right size; IntegralImage16 creates its destination).  Errors are std::runtime_error.
#ifndef BRISK_INTERNAL_IMAGE_FUNCTIONS_16_H_
#define BRISK_INTERNAL_IMAGE_FUNCTIONS_16_H_

#include <agast/wrap-opencv.h>
#include <brisk/hip-context.h>
#include <brisk_hip.h>

namespace brisk {

inline void Halfsample16(const agast::Mat& srcimg, agast::Mat& dstimg) {
  if (srcimg.type() != CV_16UC1 || srcimg.cols / 2 != dstimg.cols || srcimg.rows / 2 != dstimg.rows)
    throw std::runtime_error("Halfsample16: CV_16UC1 source, destination of half the size");
  brisk_hip_ctx* ctx = hip::DefaultContext();
  hip::Check(ctx, brisk_hip_halfsample16(ctx, reinterpret_cast<const uint16_t*>(srcimg.data), srcimg.cols, srcimg.rows,
                                         (int)(srcimg.step / 2), reinterpret_cast<uint16_t*>(dstimg.data), (int)(dstimg.step / 2)),
             "brisk_hip_halfsample16");
}

inline void Twothirdsample16(const agast::Mat& srcimg, agast::Mat& dstimg) {
  if (srcimg.type() != CV_16UC1 || (srcimg.cols / 3) * 2 != dstimg.cols || (srcimg.rows / 3) * 2 != dstimg.rows)
    throw std::runtime_error("Twothirdsample16: CV_16UC1 source, destination of two thirds the size");
  brisk_hip_ctx* ctx = hip::DefaultContext();
  hip::Check(ctx, brisk_hip_twothirdsample16(ctx, reinterpret_cast<const uint16_t*>(srcimg.data), srcimg.cols, srcimg.rows,
                                             (int)(srcimg.step / 2), reinterpret_cast<uint16_t*>(dstimg.data), (int)(dstimg.step / 2)),
             "brisk_hip_twothirdsample16");
}

inline void IntegralImage16(const agast::Mat& src, agast::Mat* dest) {
  if (!dest || src.type() != CV_16UC1) throw std::runtime_error("IntegralImage16: CV_16UC1 source, non-null destination");
  dest->create(src.rows + 1, src.cols + 1, CV_MAKETYPE(CV_32F, 1));
  brisk_hip_ctx* ctx = hip::DefaultContext();
  hip::Check(ctx, brisk_hip_integral_image16(ctx, reinterpret_cast<const uint16_t*>(src.data), src.cols, src.rows, (int)(src.step / 2),
                                             reinterpret_cast<float*>(dest->data), (int)(dest->step / 4)),
             "brisk_hip_integral_image16");
}

}  // namespace brisk
#endif  // BRISK_INTERNAL_IMAGE_FUNCTIONS_16_H_
