// fealess_opencv_adapter.hpp -- the OpenCV-typed surfaces of the reference on top of the C ABI (fealess_hip.h).
//
// Header-only; compile it only where OpenCV (3.x or later) is installed: it is the glue a CadReco maintainer would
// otherwise write by hand.  Every function keeps the reference's name, argument list and
// error behaviour and cites what it replaces:
//   * fealess_cv::readLinemod            linemod/linemod_if.h:15 (cv::Ptr<Detector> readLinemod(const std::string&))
//   * fealess_cv::writeLinemod           linemod/linemod_if.h:17 / linemod_if.cpp:49-63
//   * fealess_cv::drawResponse (x2)      linemod/linemod_if.h:19-23 / linemod_if.cpp:65-150 (host-side drawing, no GPU work)
//   * fealess_cv::Detector::match        linemod/linemod.hpp:324-327 / linemod.cpp:1356-1441
//   * fealess_cv::detection              ICP/detection.h:9-11 / ICP/detection.cpp:11-254
//   * fealess_cv::icpCloudToCloud_Ex     ICP/ICP.h:165-172 / ICP/ICP.cpp:617-809
//   * fealess_cv::depthTo3d              ICP/depth_to_3d.h:12-13 / depth_to_3d.cpp:244-269
// The reference's own names: unless FEALESS_CV_NO_REFERENCE_NAMES is defined, the header also declares them where the
// reference has them -- cup_linemod::{Detector, Match, Template, Feature}, the global readLinemod / writeLinemod /
// drawResponse / detection / icpCloudToCloud_Ex and cup_d2pc::depthTo3d -- as aliases of the fealess_cv ones, so a
// translation unit that included "linemod_if.h", "ICP.h", "detection.h" and "depth_to_3d.h" compiles against this header
// unchanged (define the macro where both sets of headers have to coexist in one translation unit).
// Build: -I include -I fealess_amd/cadreco, link libfealess_hip.so and libcadreco_hip.so (the latter for the
// linemod_templates.yml reader).  There is no OpenCV in the build image, so this header is checked for syntax against
// a minimal stand-in of <opencv2/core.hpp> (tests/dropin/opencv_stub, tests/test_dropin_cpu.py) and has not been run.
#ifndef FEALESS_OPENCV_ADAPTER_HPP
#define FEALESS_OPENCV_ADAPTER_HPP

#include <opencv2/core.hpp>
#include <opencv2/imgproc.hpp>      // cv::circle (drawResponse)

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "fealess_cadreco.h"
#include "fealess_hip.h"

namespace fealess_cv {

// cup_linemod::Match (linemod.hpp:253-281): same members, same orderings
struct Match {
  Match() : x(0), y(0), similarity(0), template_id(0) {}
  Match(int _x, int _y, float _similarity, const cv::String &_class_id, int _template_id)
      : x(_x), y(_y), similarity(_similarity), class_id(_class_id), template_id(_template_id) {}
  bool operator<(const Match &rhs) const { return similarity != rhs.similarity ? similarity > rhs.similarity : template_id < rhs.template_id; }
  bool operator==(const Match &rhs) const { return x == rhs.x && y == rhs.y && similarity == rhs.similarity && class_id == rhs.class_id; }
  int x, y;
  float similarity;
  cv::String class_id;
  int template_id;
};

// cup_linemod::Detector's read-only surface (linemod.hpp:292-391) with the bank resident in HBM.  Not thread-safe per
// object (one fl_context, calls serialised), like CObjRecoLmICP (obj_reco_lmicp.h:37-51); distinct objects are independent.
class Detector {
 public:
  ~Detector()
  {
    if (det_) fl_detector_destroy(det_);
    if (ctx_) fl_context_destroy(ctx_);
  }
  int pyramidLevels() const { return file_.pyramid_levels; }                       // linemod.hpp:349
  int getT(int pyramid_level) const { return file_.T[(size_t)pyramid_level]; }     // :344
  int numClasses() const { return (int)file_.classes.size(); }                     // :362
  int numTemplates() const { return det_ ? fl_detector_num_templates(det_) : 0; }  // :360
  std::vector<cv::String> classIds() const { return std::vector<cv::String>(class_ids_.begin(), class_ids_.end()); }   // :364 (map order)
  std::vector<float> getPoseInfo(int template_id)                                   // :339 (Q6: one global table, single-class use)
  {
    for (size_t c = 0; c < file_.classes.size(); ++c)
      if (template_id >= 0 && template_id < (int)file_.classes[c].poses.size()) return file_.classes[c].poses[(size_t)template_id];
    return std::vector<float>();
  }
  // getTemplates (linemod.hpp:357): order (GradientL0, NormalL0, GradientL1, NormalL1, ...)
  const std::vector<fealess::Template> &getTemplates(const cv::String &class_id, int template_id) const
  {
    for (size_t c = 0; c < file_.classes.size(); ++c)
      if (file_.classes[c].class_id == std::string(class_id)) return file_.classes[c].template_pyramids.at((size_t)template_id);
    CV_Error(cv::Error::StsBadArg, "class_id not found");
    static const std::vector<fealess::Template> none;    // not reached: CV_Error throws
    return none;
  }

  // Detector::match (linemod.hpp:324-327).  sources = {CV_8UC3 colour, CV_16UC1 depth} (one per modality); returns 0, or
  // -1 where the reference does (sources.size() != modalities.size()).  matches are sorted and de-duplicated as the
  // reference's std::sort + std::unique leave them (one valid outcome of the unstable sort, see fealess_hip.h).
  int match(const std::vector<cv::Mat> &sources, float threshold, std::vector<Match> &matches,
            const std::vector<cv::String> &class_ids = std::vector<cv::String>(), cv::OutputArrayOfArrays quantized_images = cv::noArray(),
            const std::vector<cv::Mat> &masks = std::vector<cv::Mat>()) const
  {
    matches.clear();
    const int M = (int)file_.modalities.size();
    if ((int)sources.size() != M) return -1;                                        // linemod.cpp:1364
    CV_Assert(masks.empty() || masks.size() == sources.size());                     // :1365
    CV_Assert(sources[0].type() == CV_8UC3 && sources[0].isContinuous());
    CV_Assert(M < 2 || (sources[1].type() == CV_16UC1 && sources[1].isContinuous() && sources[1].size() == sources[0].size()));
    const int w = sources[0].cols, h = sources[0].rows;
    ensure(w, h);
    std::vector<const char *> ids;
    for (size_t i = 0; i < class_ids.size(); ++i) ids.push_back(class_ids[i].c_str());
    check(fl_detector_set_class_filter(det_, ids.empty() ? NULL : ids.data(), (int)ids.size()));
    std::vector<cv::Mat> mk(masks.size());
    std::vector<const uint8_t *> mp(masks.size(), (const uint8_t *)NULL);
    for (size_t i = 0; i < masks.size(); ++i) {
      if (masks[i].empty()) continue;
      CV_Assert(masks[i].type() == CV_8UC1 && masks[i].size() == sources[i].size());
      mk[i] = masks[i].isContinuous() ? masks[i] : masks[i].clone();
      mp[i] = mk[i].data;
    }
    std::vector<fl_match> out(65536);
    int n = 0;
    for (;;) {
      check(fl_match_frame_masked(det_, sources[0].data, M > 1 ? (const uint16_t *)sources[1].data : NULL, masks.empty() ? NULL : mp.data(),
                                  FL_MEM_HOST, threshold, out.data(), (int)out.size(), &n));
      if (n <= (int)out.size()) break;
      out.resize((size_t)n);                                                        // the list is unbounded in the reference
    }
    matches.reserve((size_t)n);
    for (int i = 0; i < n; ++i)
      matches.push_back(Match(out[(size_t)i].x, out[(size_t)i].y, out[(size_t)i].similarity, class_ids_[(size_t)out[(size_t)i].class_idx],
                              out[(size_t)i].template_id));
    if (quantized_images.needed()) {                                                // linemod.cpp:1361-1362, 1411-1412
      const int L = file_.pyramid_levels;
      quantized_images.create(1, L * M, CV_8U);
      size_t total = 0;
      for (int l = 0; l < L; ++l) total += (size_t)(w >> l) * (size_t)(h >> l) * (size_t)M;
      std::vector<uint8_t> buf(total);
      check(fl_last_quantized(det_, buf.data()));
      size_t o = 0;
      for (int l = 0; l < L; ++l)
        for (int m = 0; m < M; ++m) {
          cv::Mat q(h >> l, w >> l, CV_8UC1, buf.data() + o);
          q.copyTo(quantized_images.getMatRef(l * M + m));
          o += (size_t)(w >> l) * (size_t)(h >> l);
        }
    }
    return 0;
  }

  const fealess::DetectorFile &file() const { return file_; }     // what writeLinemod writes back
  fl_detector *handle() const { return det_; }       // for fl_recognize_* / multi-GPU calls on the same bank
  fl_context *context() const { return ctx_; }

 private:
  friend cv::Ptr<Detector> readLinemod(const std::string &filename, int device);
  Detector() : ctx_(NULL), det_(NULL), w_(0), h_(0) {}
  static void check(int rc)
  {
    if (rc == FL_ERR_ASSERT) CV_Error(cv::Error::StsAssert, "the reference would hit a CV_Assert here (fl_last_error has the line)");
    if (rc != FL_OK) CV_Error(cv::Error::StsError, "libfealess_hip call failed");
  }
  void ensure(int w, int h) const
  {
    if (w == w_ && h == h_) return;
    check(fl_detector_finalize(det_, w, h, 1, 0));   // CV_Assert of linemod.cpp:981,1062 on sizes that do not divide by T
    w_ = w;
    h_ = h;
  }
  fealess::DetectorFile file_;
  std::vector<std::string> class_ids_;               // std::map order = class_idx of fl_match
  fl_context *ctx_;
  fl_detector *det_;
  mutable int w_, h_;
};

// readLinemod (linemod_if.h:15, linemod_if.cpp:36-47): the linemod_templates.yml written by writeLinemod, uploaded to
// `device`.  Like the reference it returns a detector with 0 classes when the file cannot be read.
inline cv::Ptr<Detector> readLinemod(const std::string &filename, int device = 0)
{
  cv::Ptr<Detector> d(new Detector());
  std::string err;
  if (!fealess::ReadLinemodCached(filename, d->file_, &err, NULL)) { d->file_ = fealess::DetectorFile(); return d; }
  const int M = (int)d->file_.modalities.size(), L = d->file_.pyramid_levels;
  CV_Assert(M >= 1 && M <= 2 && L >= 1 && (int)d->file_.T.size() >= L);
  Detector::check(fl_context_create(device, &d->ctx_));
  Detector::check(fl_detector_create(d->ctx_, M, L, d->file_.T.data(), &d->det_));
  for (size_t c = 0; c < d->file_.classes.size(); ++c) {
    const fealess::ObjectClass &oc = d->file_.classes[c];
    std::vector<fl_template> tl;
    std::vector<fl_feature> fl;
    std::vector<float> poses;
    for (size_t p = 0; p < oc.template_pyramids.size(); ++p) {
      for (size_t t = 0; t < oc.template_pyramids[p].size(); ++t) {
        const fealess::Template &tt = oc.template_pyramids[p][t];
        fl_template hdr = {tt.width, tt.height, tt.offset_x, tt.offset_y, tt.pyramid_level, (int)fl.size(), (int)tt.features.size()};
        for (size_t k = 0; k < tt.features.size(); ++k) { fl_feature f = {tt.features[k].x, tt.features[k].y, tt.features[k].label}; fl.push_back(f); }
        tl.push_back(hdr);
      }
      for (int k = 0; k < 13; ++k) poses.push_back(p < oc.poses.size() && k < (int)oc.poses[p].size() ? oc.poses[p][(size_t)k] : 0.f);
    }
    Detector::check(fl_detector_add_class(d->det_, oc.class_id.c_str(), (int)oc.template_pyramids.size(), tl.data(), fl.data(), (int)fl.size(),
                                          poses.data()));
    d->class_ids_.push_back(oc.class_id);
  }
  std::sort(d->class_ids_.begin(), d->class_ids_.end());
  return d;
}

// writeLinemod (linemod_if.h:17, linemod_if.cpp:49-63): Detector::write + writeClass for every class into one
// FileStorage YAML (linemod.cpp:1696-1794) -- the file readLinemod reads
inline void writeLinemod(const cv::Ptr<Detector> &detector, const std::string &filename)
{
  if (!detector || !fealess::WriteLinemod(detector->file(), filename)) CV_Error(cv::Error::StsError, "writeLinemod: cannot write the file");
}

// drawResponse (linemod_if.h:19-20, linemod_if.cpp:65-88): one circle of radius T / 2 per feature, the colour by modality
inline void drawResponse(const std::vector<fealess::Template> &templates, int num_modalities, cv::Mat &dst, cv::Point offset, int T)
{
  static const cv::Scalar COLORS[5] = {cv::Scalar(0, 140, 255), cv::Scalar(0, 255, 0), cv::Scalar(0, 255, 255), cv::Scalar(0, 140, 255),
                                       cv::Scalar(0, 0, 255)};                       // CV_RGB(r, g, b) = Scalar(b, g, r)
  for (int m = 0; m < num_modalities; ++m)
    for (size_t i = 0; i < templates[(size_t)m].features.size(); ++i) {
      const fealess::Feature &f = templates[(size_t)m].features[i];
      cv::circle(dst, cv::Point(f.x + offset.x, f.y + offset.y), T / 2, COLORS[m], 2);
    }
}
// drawResponse with the rendered template pasted in first (linemod_if.h:22-23, linemod_if.cpp:90-150): the bounding box of
// current_template's non-black pixels (rows in min_x .. max_x, columns in min_y .. max_y, the reference's naming), grown by
// one at its far edges, is copied to dst at `offset` wherever it is non-black; then the circles as above
inline void drawResponse(const std::vector<fealess::Template> &templates, int num_modalities, cv::Mat &dst, cv::Point offset, int T,
                         cv::Mat current_template)
{
  int min_x = 1000, min_y = 1000, max_x = 0, max_y = 0;
  for (int i = 0; i < current_template.rows; ++i)
    for (int j = 0; j < current_template.cols; ++j) {
      const cv::Vec3b &p = current_template.at<cv::Vec3b>(i, j);
      if (p[0] != 0 || p[1] != 0 || p[2] != 0) {
        min_x = std::min(min_x, i); max_x = std::max(max_x, i);
        min_y = std::min(min_y, j); max_y = std::max(max_y, j);
      }
    }
  if (max_x < current_template.rows - 1) max_x += 1;
  if (max_y < current_template.cols - 1) max_y += 1;
  for (int i = min_x; i < max_x; ++i)
    for (int j = min_y; j < max_y; ++j) {
      const cv::Vec3b &p = current_template.at<cv::Vec3b>(i, j);
      if (p[0] != 0 || p[1] != 0 || p[2] != 0) dst.at<cv::Vec3b>(i - min_x + offset.y, j - min_y + offset.x) = p;
    }
  drawResponse(templates, num_modalities, dst, offset, T);
}

// one context for the free functions below (the reference's are stateless)
inline fl_context *default_context()
{
  static fl_context *ctx = NULL;
  if (!ctx && fl_context_create(0, &ctx) != FL_OK) CV_Error(cv::Error::StsError, "no HIP device (there is no CPU path)");
  return ctx;
}

// detection() (ICP/detection.h:9-11): both depth images CV_16UC1 in millimetres and of equal size; d_match is unused by the
// reference as well (detection.cpp:11-254).  A crop rectangle that leaves the image throws, as the reference's cv::Mat ROI
// does (detection.cpp:43-44).
inline void detection(cv::Mat depImg_model_raw, cv::Mat depImg_ref_raw, TCamIntrinsicParam tCamIntrinsic, const cv::Rect_<int> rect_model_raw,
                      cv::Rect_<int> rect_ref_raw, int icp_it_thr, float dist_mean_thr, float dist_diff_thr, cv::Matx33f r_match,
                      cv::Vec3f t_match, float /*d_match*/, cv::Vec3f &T_final, cv::Matx33f &R_final, int icp_mode = FL_ICP_PARITY)
{
  CV_Assert(depImg_model_raw.type() == CV_16UC1 && depImg_ref_raw.type() == CV_16UC1 && depImg_model_raw.size() == depImg_ref_raw.size());
  cv::Mat m = depImg_model_raw.isContinuous() ? depImg_model_raw : depImg_model_raw.clone();
  cv::Mat s = depImg_ref_raw.isContinuous() ? depImg_ref_raw : depImg_ref_raw.clone();
  fl_intrinsics K = {s.cols, s.rows, tCamIntrinsic.dFx, tCamIntrinsic.dFy, tCamIntrinsic.dCx, tCamIntrinsic.dCy};
  const int rm[4] = {rect_model_raw.x, rect_model_raw.y, rect_model_raw.width, rect_model_raw.height};
  const int rr[4] = {rect_ref_raw.x, rect_ref_raw.y, rect_ref_raw.width, rect_ref_raw.height};
  fl_detection_result res;
  const int rc = fl_detection(default_context(), (const uint16_t *)m.data, (const uint16_t *)s.data, s.cols, s.rows, &K, rm, rr, icp_it_thr,
                              dist_mean_thr, dist_diff_thr, r_match.val, t_match.val, icp_mode, FL_MEM_HOST, &res);
  if (rc == FL_ERR_ASSERT) CV_Error(cv::Error::StsAssert, "crop rectangle leaves the image (cv::Mat ROI, detection.cpp:43-44)");
  if (rc != FL_OK) CV_Error(cv::Error::StsError, fl_last_error(default_context()));
  for (int i = 0; i < 9; ++i) R_final.val[i] = res.R_final[i];
  for (int i = 0; i < 3; ++i) T_final.val[i] = res.T_final[i];
}

// icpCloudToCloud_Ex (ICP/ICP.h:165-172): returns dist_mean (-1 with fewer than 3 points)
inline float icpCloudToCloud_Ex(const std::vector<cv::Vec3f> &pts_ref, const std::vector<cv::Vec3f> &pts_model, cv::Matx33f &R, cv::Vec3f &T,
                                float &px_ratio_match, int icp_it_th = 4, const float dist_mean_thr = 0.0f, const float dist_diff_thr = 0.0f,
                                int icp_mode = FL_ICP_PARITY)
{
  fl_icp_result res;
  const int rc = fl_icp(default_context(), pts_ref.empty() ? NULL : pts_ref[0].val, (int)pts_ref.size(), pts_model.empty() ? NULL : pts_model[0].val,
                        (int)pts_model.size(), icp_it_th, dist_mean_thr, dist_diff_thr, icp_mode, FL_MEM_HOST, &res);
  if (rc != FL_OK) CV_Error(cv::Error::StsError, fl_last_error(default_context()));
  for (int i = 0; i < 9; ++i) R.val[i] = res.R[i];
  for (int i = 0; i < 3; ++i) T.val[i] = res.T[i];
  px_ratio_match = res.px_ratio;
  return res.dist_mean;
}

// cup_d2pc::depthTo3d (depth_to_3d.h:12-13) for the case the path uses: CV_16UC1 depth in millimetres, no mask, 3x3 K of
// any float type -> CV_32FC3 points in metres, NaN where the depth is 0 (depth_to_3d.cpp:99-137, 244-269)
inline void depthTo3d(cv::InputArray depth_in, cv::InputArray K_in, cv::OutputArray points3d_out)
{
  cv::Mat depth = depth_in.getMat(), K = K_in.getMat();
  CV_Assert(depth.type() == CV_16UC1 && K.rows == 3 && K.cols == 3);
  cv::Mat Kd;
  K.convertTo(Kd, CV_64F);
  cv::Mat d = depth.isContinuous() ? depth : depth.clone();
  points3d_out.create(d.rows, d.cols, CV_32FC3);
  cv::Mat pts = points3d_out.getMat();
  const int rc = fl_depth_to_3d(default_context(), (const uint16_t *)d.data, d.cols, d.rows, Kd.at<double>(0, 0), Kd.at<double>(1, 1), Kd.at<double>(0, 2),
                                Kd.at<double>(1, 2), (float *)pts.data, FL_MEM_HOST);
  if (rc != FL_OK) CV_Error(cv::Error::StsError, fl_last_error(default_context()));
}

}  // namespace fealess_cv

#ifndef FEALESS_CV_NO_REFERENCE_NAMES
// the names a CadReco translation unit already uses (linemod.hpp, linemod_if.h, ICP.h, detection.h, depth_to_3d.h)
namespace cup_linemod {
using fealess_cv::Detector;
using fealess_cv::Match;
typedef fealess::Template Template;
typedef fealess::Feature Feature;
}  // namespace cup_linemod
using fealess_cv::readLinemod;
using fealess_cv::writeLinemod;
using fealess_cv::drawResponse;
using fealess_cv::detection;
using fealess_cv::icpCloudToCloud_Ex;
namespace cup_d2pc {
using fealess_cv::depthTo3d;
}  // namespace cup_d2pc
#endif
#endif  // FEALESS_OPENCV_ADAPTER_HPP
