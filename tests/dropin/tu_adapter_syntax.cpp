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
// the same calls under the reference's own names (linemod_if.h:15-23, linemod.hpp, ICP.h:165, detection.h:9, depth_to_3d.h:10-13)
void use_reference_names(const std::string &yml, cv::Mat bgr, cv::Mat depth, cv::Mat model, TCamIntrinsicParam K)
{
  cv::Ptr<cup_linemod::Detector> det = readLinemod(yml);
  writeLinemod(det, yml + ".copy");
  std::vector<cup_linemod::Match> matches;
  std::vector<cv::Mat> src;
  src.push_back(bgr);
  src.push_back(depth);
  det->match(src, 75.0f, matches);
  const std::vector<cup_linemod::Template> &templates = det->getTemplates(matches[0].class_id, matches[0].template_id);
  cup_linemod::Feature f = templates[0].features[0];
  (void)f;
  drawResponse(templates, 2, bgr, cv::Point(matches[0].x, matches[0].y), det->getT(0));
  drawResponse(templates, 2, bgr, cv::Point(matches[0].x, matches[0].y), det->getT(0), model);
  cv::Matx33f R, r_match = cv::Matx33f();
  cv::Vec3f T, t_match = cv::Vec3f();
  cv::Rect_<int> rm = {0, 0, 8, 8}, rr = {0, 0, 8, 8};
  detection(model, depth, K, rm, rr, 10, 0.5f, 0.01f, r_match, t_match, 0.f, T, R);
  std::vector<cv::Vec3f> a, b;
  float px = 0.f;
  (void)icpCloudToCloud_Ex(a, b, R, T, px);
  cv::Mat Kmat, pts;
  cup_d2pc::depthTo3d(depth, Kmat, pts);
}
