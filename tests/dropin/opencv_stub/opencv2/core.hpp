// Minimal stand-in of <opencv2/core.hpp> for ONE purpose: a syntax check (g++ -fsyntax-only) of
// include/fealess_opencv_adapter.hpp in an image without OpenCV (tests/test_dropin_cpu.py).  Declarations only, no
// behaviour, nothing here is linked or run; it is NOT a build of the reference and is not part of the product.
#ifndef FEALESS_TEST_OPENCV_STUB
#define FEALESS_TEST_OPENCV_STUB
#include <memory>
#include <string>
#include <vector>
#define CV_8U 0
#define CV_64F 6
#define CV_8UC1 0
#define CV_8UC3 16
#define CV_16UC1 2
#define CV_32FC3 21
namespace cv {
typedef std::string String;
template <typename T> using Ptr = std::shared_ptr<T>;
struct Size { int width, height; bool operator==(const Size &o) const; };
template <typename T> struct Rect_ { T x, y, width, height; };
struct Matx33f { float val[9]; };
struct Vec3f { float val[3]; };
struct Vec3b { unsigned char val[3]; unsigned char &operator[](int i); const unsigned char &operator[](int i) const; };
struct Point { int x, y; Point(); Point(int x_, int y_); };
struct Scalar { double val[4]; Scalar(); Scalar(double a, double b, double c); };
class Mat {
 public:
  Mat();
  Mat(int rows, int cols, int type, void *data);
  int rows, cols;
  unsigned char *data;
  int type() const;
  bool isContinuous() const;
  bool empty() const;
  Size size() const;
  Mat clone() const;
  void copyTo(Mat &dst) const;
  void convertTo(Mat &dst, int rtype) const;
  template <typename T> T &at(int r, int c);
};
class _InputArray { public: Mat getMat() const; _InputArray(const Mat &); };
class _OutputArray {
 public:
  _OutputArray(Mat &);
  _OutputArray(std::vector<Mat> &);
  _OutputArray();
  bool needed() const;
  void create(int rows, int cols, int type) const;
  Mat getMat() const;
  Mat &getMatRef(int i) const;
};
typedef const _InputArray &InputArray;
typedef const _OutputArray &OutputArray;
typedef const _OutputArray &OutputArrayOfArrays;
const _OutputArray &noArray();
namespace Error { enum { StsError = -2, StsBadArg = -5, StsAssert = -215 }; }
[[noreturn]] void error(int code, const String &msg, const char *func, const char *file, int line);
}  // namespace cv
#define CV_Error(code, msg) cv::error(code, msg, __func__, __FILE__, __LINE__)
#define CV_Assert(expr) do { if (!(expr)) cv::error(cv::Error::StsAssert, #expr, __func__, __FILE__, __LINE__); } while (0)
#endif
