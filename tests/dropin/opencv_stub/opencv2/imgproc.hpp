// Minimal stand-in of <opencv2/imgproc.hpp> for the syntax check of include/fealess_opencv_adapter.hpp (see core.hpp here):
// a declaration only, nothing is linked or run.
#ifndef FEALESS_TEST_OPENCV_STUB_IMGPROC
#define FEALESS_TEST_OPENCV_STUB_IMGPROC
#include "core.hpp"
namespace cv {
void circle(Mat &img, Point center, int radius, const Scalar &color, int thickness = 1, int lineType = 8, int shift = 0);
}
#endif
