#include <opencv2/features2d/features2d.hpp>
