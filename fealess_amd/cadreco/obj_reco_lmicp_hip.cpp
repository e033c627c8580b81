// obj_reco_lmicp_hip.cpp -- the CadReco facade (CObjRecoCAD / CObjRecoLmICP) on top of the C ABI
// of libfealess_hip.so.  Mirrors, call for call, CadReco/obj_reco_lmicp.cpp:47-259 and
// CadReco/obj_reco_temp.cpp:6-35 of the reference: same entry points, argument meaning, return
// codes and defaults; the OpenCV-typed private members are gone (frames stay where the caller
// put them until the ABI uploads them to HBM).
#include "fealess_cadreco.h"
#include "../../include/fealess_hip.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <sstream>

#define PROC_IMG_WIDTH 640      // obj_reco_lmicp.cpp:6

namespace {
int check_image_u8(const TImageU &t) { return t.dTimestamp >= 0 && t.nHeight > 0 && t.nWidth > 0 && t.pData; }     // CheckTImage :32-36
int check_image_u16(const TImageU16 &t) { return t.dTimestamp >= 0 && t.nHeight > 0 && t.nWidth > 0 && t.pData; }
}  // namespace

class CObjRecoLmICPHip : public CObjRecoCAD {
 public:
  CObjRecoLmICPHip()
  {
    // constructor defaults of obj_reco_lmicp.cpp:47-56
    m_params.matching_threshold = 75.0f;
    m_params.icp_it_thr = 10;
    m_params.dist_mean_thr = 0.5f;
    m_params.dist_diff_thr = 0.01f;
    m_params.icp_mode = FL_ICP_PARITY;
    if (fl_context_create(0, &m_ctx) != FL_OK) m_ctx = nullptr;    // no GPU: every call fails, no CPU path
  }
  ~CObjRecoLmICPHip() override
  {
    if (m_det) fl_detector_destroy(m_det);
    if (m_ctx) fl_context_destroy(m_ctx);
  }
  int Train(const string &, const TScanPackage &, const TTrainParam &) override { return 0; }   // stub in the reference too (:62-65)
  int ClearObj() override { return 0; }                                                          // :76-79
  int SetROI(const TImageU &) override { return 0; }                                             // :81-84
  int SetAdvancedParam(const AdvancedParam &) override { return 0; }                             // :206-209
  int GetAdvancedParam(const string &, void *) override { return 0; }                            // :211-214

  // AddObj (:67-74): loads <dir>/linemod_templates.yml; the depth renders the reference re-reads
  // from <dir>/depth/<id>.png on every frame (:156-157) are uploaded to HBM here, once.
  int AddObj(const string str_feature_path) override
  {
    if (!m_ctx) return (int)ERROR_UNKNOW;
    fealess::DetectorFile df;
    std::string err;
    if (!fealess::ReadLinemodCached(str_feature_path + "/linemod_templates.yml", df, &err, nullptr) || df.classes.empty())
      return (int)ERROR_OPEN_FILE_FAILED;                                   // numClasses() == 0 (:71-72)
    if (m_det) { fl_detector_destroy(m_det); m_det = nullptr; }
    const int M = (int)df.modalities.size(), L = df.pyramid_levels;
    // the file is untrusted input: the C ABI reads T[0..L) and one 13-float pose per pyramid
    if (M < 1 || M > 2 || L < 1 || L > 4 || (int)df.T.size() < L) return (int)ERROR_VERSION_MISMATCH;
    for (auto &c : df.classes)
      if (c.poses.size() != c.template_pyramids.size()) return (int)ERROR_VERSION_MISMATCH;
    if (fl_detector_create(m_ctx, M, L, df.T.data(), &m_det) != FL_OK) return (int)ERROR_INVALID_PARAM;
    m_class_ids.clear();
    std::vector<std::vector<unsigned short> > depth_banks;
    for (auto &c : df.classes) {
      std::vector<fl_template> tl;
      std::vector<fl_feature> fl;
      std::vector<float> poses;
      for (size_t p = 0; p < c.template_pyramids.size(); ++p) {
        for (auto &t : c.template_pyramids[p]) {
          fl_template h = {t.width, t.height, t.offset_x, t.offset_y, t.pyramid_level, (int)fl.size(), (int)t.features.size()};
          for (auto &f : t.features) fl.push_back(fl_feature{f.x, f.y, f.label});
          tl.push_back(h);
        }
        for (int k = 0; k < 13; ++k) poses.push_back(k < (int)c.poses[p].size() ? c.poses[p][k] : 0.f);
      }
      if ((int)tl.size() != (int)c.template_pyramids.size() * L * M) return (int)ERROR_VERSION_MISMATCH;
      if (fl_detector_add_class(m_det, c.class_id.c_str(), (int)c.template_pyramids.size(), tl.data(), fl.data(), (int)fl.size(),
                                poses.data()) != FL_OK)
        return (int)ERROR_INVALID_PARAM;
      m_class_ids.push_back(c.class_id);
    }
    std::sort(m_class_ids.begin(), m_class_ids.end());                      // class_idx = std::map order
    // depth renders (all classes share one directory in the reference: single-class use, Q6)
    for (size_t ci = 0; ci < df.classes.size(); ++ci) {
      int cidx = (int)(std::find(m_class_ids.begin(), m_class_ids.end(), df.classes[ci].class_id) - m_class_ids.begin());
      const int n = (int)df.classes[ci].template_pyramids.size();
      for (int p = 0; p < n; ++p) {
        std::ostringstream fn;
        fn << str_feature_path << "/depth/" << p << ".png";
        std::vector<unsigned short> px;
        int w = 0, h = 0;
        if (!fealess::ReadPng16(fn.str(), px, w, h, &err)) continue;        // imread failure surfaces at Recognition time
        if (fl_detector_set_model_depths(m_det, cidx, p, 1, px.data(), w, h, FL_MEM_HOST) != FL_OK) return (int)ERROR_INVALID_PARAM;
      }
    }
    m_w = m_h = 0;
    m_path = str_feature_path;
    return 0;
  }

  int Recognition(const TImageU &tRGB, const TImageU16 &tDepth, const TCamIntrinsicParam &K, vector<TObjRecoResult> &vtResult) override
  {
    std::vector<std::vector<TObjRecoResult> > out;
    int rc = Batch(1, &tRGB, &tDepth, K, out);
    vtResult.clear();
    if (rc == 0 && !out.empty()) vtResult = out[0];
    return rc;
  }

  // frame_rc (optional): what Recognition() would have returned for each frame; the return value is the first non-zero
  // of them, and the frames that succeeded keep their results in `out` either way
  int Batch(int n, const TImageU *rgb, const TImageU16 *depth, const TCamIntrinsicParam &K, std::vector<std::vector<TObjRecoResult> > &out,
            std::vector<int> *frame_rc = nullptr)
  {
    out.clear();
    if (frame_rc) frame_rc->assign(n > 0 ? n : 0, (int)ERROR_INVALID_PARAM);
    if (!m_ctx || !m_det || n <= 0) return (int)ERROR_INVALID_PARAM;
    // PrepareInputData (:216-259)
    for (int i = 0; i < n; ++i) {
      if (!check_image_u8(rgb[i]) || !check_image_u16(depth[i])) return (int)ERROR_INVALID_PARAM;
      if (rgb[i].nHeight != K.nHeight || rgb[i].nWidth != K.nWidth || depth[i].nHeight != K.nHeight || depth[i].nWidth != K.nWidth)
        return (int)ERROR_INVALID_PARAM;
    }
    // zoom to width 640 (:229-249): w = 640, h = H * 640 / W (integer), cv::resize(INTER_LINEAR) of both images
    const int w = PROC_IMG_WIDTH, h = K.nHeight * PROC_IMG_WIDTH / K.nWidth;
    if (h <= 0) return (int)ERROR_INVALID_PARAM;
    std::vector<const uint8_t *> bp(n);
    std::vector<const uint16_t *> dp(n);
    std::vector<std::vector<uint8_t> > zb;
    std::vector<std::vector<uint16_t> > zd;
    const bool zoom = K.nWidth != w;
    // the single-hypothesis path zooms on the device (fl_recognize_batch_zoom); only the multi-hypothesis extension still
    // takes the zoomed frames through host vectors
    if (zoom && m_topk > 1) {
      zb.resize(n);
      zd.resize(n);
      for (int i = 0; i < n; ++i) {
        zb[i].resize((size_t)w * h * 3);
        zd[i].resize((size_t)w * h);
        if (fl_resize_linear_bgr8(m_ctx, rgb[i].pData, K.nWidth, K.nHeight, zb[i].data(), w, h, FL_MEM_HOST) != FL_OK ||
            fl_resize_linear_u16(m_ctx, depth[i].pData, K.nWidth, K.nHeight, zd[i].data(), w, h, FL_MEM_HOST) != FL_OK) {
          fprintf(stderr, "[fealess_hip] %s\n", fl_last_error(m_ctx));
          return (int)ERROR_INVALID_PARAM;
        }
        bp[i] = zb[i].data();
        dp[i] = zd[i].data();
      }
    } else {
      for (int i = 0; i < n; ++i) { bp[i] = rgb[i].pData; dp[i] = depth[i].pData; }     // zoomed on the device when `zoom`
    }
    if (m_w != w || m_h != h || n > m_batch) {
      if (fl_detector_finalize(m_det, w, h, n > m_batch ? n : m_batch, 0) != FL_OK) {
        fprintf(stderr, "[fealess_hip] %s\n", fl_last_error(m_ctx));
        return (int)ERROR_INVALID_PARAM;
      }
      m_w = w;
      m_h = h;
      if (n > m_batch) m_batch = n;
    }
    // NB: like the reference (:190) detection() gets the caller's UN-zoomed intrinsics together with the zoomed
    // depth image (the zoomed copy only feeds SetCamIntrinsic, :236-244); identical when the input is 640 wide
    fl_intrinsics k = {w, h, K.dFx, K.dFy, K.dCx, K.dCy};
    out.resize(n);
    if (m_topk > 1) {
      // opt-in extension (CadRecoSetMultiHypothesis): the first m_topk matches of every frame are refined and
      // nonMaximumSuppression (ICP/NMS.cpp:6-40) keeps one hypothesis per object position, best first
      std::vector<fl_recognition_result> res((size_t)n * m_topk);
      std::vector<int> cnt(n), win(m_topk);
      if (fl_recognize_batch_topk(m_det, n, bp.data(), dp.data(), FL_MEM_HOST, &k, &m_params, m_topk, res.data(), cnt.data()) != FL_OK) {
        fprintf(stderr, "[fealess_hip] %s\n", fl_last_error(m_ctx));
        return (int)ERROR_INVALID_PARAM;
      }
      for (int i = 0; i < n; ++i) {
        const fl_recognition_result *r = res.data() + (size_t)i * m_topk;
        if (cnt[i] > 0 && r[0].status != FL_OK && r[0].status != FL_ERR_ASSERT) return (int)ERROR_INVALID_PARAM;
        if (frame_rc) (*frame_rc)[i] = 0;
        int nw = 0;
        if (fl_nms(r, cnt[i], m_nms_dist, win.data(), &nw) != FL_OK) return (int)ERROR_INVALID_PARAM;
        for (int g = 0; g < nw; ++g) {
          const fl_recognition_result &h = r[win[g]];
          if (!h.found) continue;
          TObjRecoResult o;
          o.strObjTag = m_class_ids[h.best.class_idx];
          memcpy(o.tWorld2Cam, h.pose, sizeof(o.tWorld2Cam));
          out[i].push_back(o);
        }
      }
      return 0;
    }
    std::vector<fl_recognition_result> res(n);
    const int rc = zoom ? fl_recognize_batch_zoom(m_det, n, bp.data(), dp.data(), K.nWidth, K.nHeight, FL_MEM_HOST, &k, &m_params, res.data())
                        : fl_recognize_batch(m_det, n, bp.data(), dp.data(), FL_MEM_HOST, &k, &m_params, res.data());
    if (rc != FL_OK) {
      fprintf(stderr, "[fealess_hip] %s\n", fl_last_error(m_ctx));
      return (int)ERROR_INVALID_PARAM;
    }
    int first_rc = 0;
    for (int i = 0; i < n; ++i) {
      // match() returned -1 / ROI assert / candidate buffers at a hard cap: that frame fails, the others keep their results
      const int frc = res[i].status != FL_OK ? (int)ERROR_INVALID_PARAM : 0;
      if (frame_rc) (*frame_rc)[i] = frc;
      if (frc) { if (!first_rc) first_rc = frc; continue; }
      if (!res[i].found) continue;                                          // vtResult stays empty, return 0 (:106-109)
      TObjRecoResult r;
      r.strObjTag = m_class_ids[res[i].best.class_idx];                     // cur_match.class_id (:112)
      memcpy(r.tWorld2Cam, res[i].pose, sizeof(r.tWorld2Cam));              // Convert() (:20-30,197)
      out[i].push_back(r);
    }
    return first_rc;
  }

  fl_recognition_params m_params;
  int m_topk = 1;            // > 1: multi-hypothesis mode (CadRecoSetMultiHypothesis)
  float m_nms_dist = 20.0f;  // th_obj_dist of nonMaximumSuppression, mm

 private:
  fl_context *m_ctx = nullptr;
  fl_detector *m_det = nullptr;
  std::vector<std::string> m_class_ids;
  std::string m_path;
  int m_w = 0, m_h = 0, m_batch = 1;
};

// ---- factory (CadReco/obj_reco_temp.cpp:6-35) -------------------------------------------------
#define LIB_VERSION "3.1.1-hip"
string CObjRecoCAD::GetVersion()
{
  std::stringstream s;
  s << "CAD-based 3D Object Recognition (MI355X / HIP build). Version " << LIB_VERSION << " Compile Time: " << __DATE__ << " " << __TIME__;
  return s.str();
}

CObjRecoCAD *CObjRecoCAD::Create(EObjRecoType eType)
{
  switch (eType) {
    case EObjReco_LmICP: return new CObjRecoLmICPHip();
    default: break;                // EObjReco_FEATURE / BB8 / PoseNet: unsupported in the reference as well
  }
  return nullptr;
}

void CObjRecoCAD::Destroy(CObjRecoCAD *pHandle) { delete pHandle; }

int CadRecoRecognitionBatch(CObjRecoCAD *handle, int n_frames, const TImageU *rgb, const TImageU16 *depth,
                            const TCamIntrinsicParam &K, std::vector<std::vector<TObjRecoResult> > &out, std::vector<int> *frame_rc)
{
  CObjRecoLmICPHip *h = dynamic_cast<CObjRecoLmICPHip *>(handle);
  if (!h) return (int)ERROR_INVALID_PARAM;
  return h->Batch(n_frames, rgb, depth, K, out, frame_rc);
}

int CadRecoSetMultiHypothesis(CObjRecoCAD *handle, int k, float nms_dist_mm)
{
  CObjRecoLmICPHip *h = dynamic_cast<CObjRecoLmICPHip *>(handle);
  if (!h || k < 1 || k > 1024 || !(nms_dist_mm >= 0.f)) return (int)ERROR_INVALID_PARAM;
  h->m_topk = k;
  h->m_nms_dist = nms_dist_mm;
  return 0;
}

// ---- flat C shim so that the pytest harness (ctypes) can drive the C++ facade -------------------
extern "C" {
int cadreco_set_multi_hypothesis(void *h, int k, float nms_dist_mm) { return CadRecoSetMultiHypothesis((CObjRecoCAD *)h, k, nms_dist_mm); }
// Recognition() returning every result: poses16 receives min(*n_results, cap) 4x4 matrices
int cadreco_recognition_all(void *h, const unsigned char *bgr, const unsigned short *depth, int w, int h_, double ts, int kw, int kh,
                            double fx, double fy, double cx, double cy, int *n_results, float *poses16, int cap)
{
  TImageU rgb = {ts, (unsigned char *)bgr, w, h_};
  TImageU16 dep = {ts, (unsigned short *)depth, w, h_};
  TCamIntrinsicParam K;
  K.nWidth = kw; K.nHeight = kh; K.dFx = fx; K.dFy = fy; K.dCx = cx; K.dCy = cy;
  vector<TObjRecoResult> out;
  const int rc = ((CObjRecoCAD *)h)->Recognition(rgb, dep, K, out);
  *n_results = (int)out.size();
  for (int i = 0; i < (int)out.size() && i < cap; ++i) memcpy(poses16 + 16 * i, out[i].tWorld2Cam, 16 * sizeof(float));
  return rc;
}
void *cadreco_create(int type) { return CObjRecoCAD::Create((CObjRecoCAD::EObjRecoType)type); }
void cadreco_destroy(void *h) { CObjRecoCAD::Destroy((CObjRecoCAD *)h); }
int cadreco_add_obj(void *h, const char *dir) { return ((CObjRecoCAD *)h)->AddObj(dir); }
int cadreco_set_params(void *h, float thr, int it, float dmean, float ddiff, int mode)
{
  CObjRecoLmICPHip *p = dynamic_cast<CObjRecoLmICPHip *>((CObjRecoCAD *)h);
  if (!p) return -1;
  p->m_params.matching_threshold = thr;
  p->m_params.icp_it_thr = it;
  p->m_params.dist_mean_thr = dmean;
  p->m_params.dist_diff_thr = ddiff;
  p->m_params.icp_mode = mode;
  return 0;
}
// returns Recognition()'s code; *n_results = vtResult.size(); pose16 / tag filled for result 0
int cadreco_recognition(void *h, const unsigned char *bgr, const unsigned short *depth, int w, int h_, double ts, int kw, int kh,
                        double fx, double fy, double cx, double cy, int *n_results, float *pose16, char *tag, int tag_cap)
{
  TImageU rgb = {ts, (unsigned char *)bgr, w, h_};
  TImageU16 d = {ts, (unsigned short *)depth, w, h_};
  TCamIntrinsicParam K;
  K.nWidth = kw; K.nHeight = kh; K.dFx = fx; K.dFy = fy; K.dCx = cx; K.dCy = cy;
  std::vector<TObjRecoResult> out;
  int rc = ((CObjRecoCAD *)h)->Recognition(rgb, d, K, out);
  *n_results = (int)out.size();
  if (!out.empty()) {
    memcpy(pose16, out[0].tWorld2Cam, 16 * sizeof(float));
    snprintf(tag, tag_cap, "%s", out[0].strObjTag.c_str());
  }
  return rc;
}
int cadreco_read_linemod_cached(const char *path, int *from_cache, int *n_templates, int *n_features)
{
  fealess::DetectorFile df;
  std::string err;
  bool fc = false;
  if (!fealess::ReadLinemodCached(path, df, &err, &fc)) return -1;
  *from_cache = fc ? 1 : 0;
  *n_templates = 0;
  *n_features = 0;
  for (auto &c : df.classes)
    for (auto &p : c.template_pyramids) {
      ++*n_templates;
      for (auto &t : p) *n_features += (int)t.features.size();
    }
  return 0;
}
int cadreco_read_linemod(const char *path, int *levels, int *n_classes, int *n_templates, int *n_features)
{
  fealess::DetectorFile df;
  std::string err;
  if (!fealess::ReadLinemod(path, df, &err)) return -1;
  *levels = df.pyramid_levels;
  *n_classes = (int)df.classes.size();
  int nt = 0, nf = 0;
  for (auto &c : df.classes)
    for (auto &p : c.template_pyramids) { ++nt; for (auto &t : p) nf += (int)t.features.size(); }
  *n_templates = nt;
  *n_features = nf;
  return 0;
}
int cadreco_read_png16(const char *path, unsigned short *out, int cap, int *w, int *h)
{
  std::vector<unsigned short> px;
  std::string err;
  if (!fealess::ReadPng16(path, px, *w, *h, &err)) return -1;
  if ((int)px.size() > cap) return -2;
  memcpy(out, px.data(), px.size() * 2);
  return 0;
}
const char *cadreco_version() { static std::string v = CObjRecoCAD::GetVersion(); return v.c_str(); }
}
