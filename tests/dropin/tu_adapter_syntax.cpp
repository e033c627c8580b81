// syntax check of the OpenCV-typed adapter header against the stand-in <opencv2/core.hpp>; instantiates every entry point
#include "fealess_opencv_adapter.hpp"
void use_all(const std::string &yml, cv::Mat bgr, cv::Mat depth, cv::Mat model, TCamIntrinsicParam K)
{
  cv::Ptr<fealess_cv::Detector> det = fealess_cv::readLinemod(yml);
  std::vector<fealess_cv::Match> matches;
  std::vector<cv::Mat> src, quant;
  src.push_back(bgr);
  src.push_back(depth);
  det->match(src, 75.0f, matches);
  det->match(src, 75.0f, matches, det->classIds(), quant, std::vector<cv::Mat>(2));
  (void)det->numTemplates(); (void)det->numClasses(); (void)det->pyramidLevels(); (void)det->getT(0); (void)det->getPoseInfo(0);
  (void)det->getTemplates("obj", 0);
  cv::Matx33f R, r_match = cv::Matx33f();
  cv::Vec3f T, t_match = cv::Vec3f();
  cv::Rect_<int> rm = {0, 0, 8, 8}, rr = {0, 0, 8, 8};
  fealess_cv::detection(model, depth, K, rm, rr, 10, 0.5f, 0.01f, r_match, t_match, 0.f, T, R);
  std::vector<cv::Vec3f> a, b;
  float px = 0.f;
  (void)fealess_cv::icpCloudToCloud_Ex(a, b, R, T, px);
  cv::Mat Kmat, pts;
  fealess_cv::depthTo3d(depth, Kmat, pts);
}
