// fealess_cadreco.h -- C++ host surface of the MI355X implementation: the CadReco facade
// (reference: CadReco/obj_reco_temp.h:6-30, CadReco/lotus_common.h, CadReco/obj_reco_lmicp.h) and
// the OpenCV-free parts of linemod/linemod_if.h, re-declared without any OpenCV type so that a
// CadReco caller links against libcadreco_hip.so instead of the reference's static libraries.
//
// Types keep the reference's names, member order and meaning, so they are layout-compatible
// with code compiled against the reference's own lotus_common.h / obj_reco_temp.h (that code may
// keep including its own headers).
#ifndef FEALESS_CADRECO_H
#define FEALESS_CADRECO_H

#include <string>
#include <vector>

#ifndef __COMMON_H__            // the reference's lotus_common.h guard: do not redefine its types
#define SUCCESS 0
#define ERROR_INVALID_PARAM    0x80000001
#define ERROR_OPEN_FILE_FAILED 0x80000002
#define ERROR_VERSION_MISMATCH 0x80000003
#define ERROR_NEW_FAILED       0x80000004
#define ERROR_UNKNOW           0x80000005

using std::string;
using std::vector;

template <typename T>
struct TImage {                 // borrowed pixel buffer; BGR8 interleaved for the colour frame
  double dTimestamp;            // ms; negative = invalid
  T *pData;
  int nWidth;
  int nHeight;
};
typedef TImage<unsigned char> TImageU;
typedef TImage<unsigned short int> TImageU16;
typedef TImage<float> TImageF;

struct TCamIntrinsicParam {     // pinhole K = [fx 0 cx; 0 fy cy; 0 0 1]
  int nWidth;
  int nHeight;
  double dFx;
  double dFy;
  double dCx;
  double dCy;
  vector<double> vdDistCoeff;
};

typedef float Mat4x4F[16];

struct TScanFrame {
  TImageU tGrayImg;
  TImageU tMask;
  Mat4x4F tWorld2Cam;
  TImageF tDepthImg;
};

struct TScanPackage {
  string strObjTag;
  Mat4x4F tGLPrjMatrix;
  vector<float> bounding_box;
  vector<TScanFrame> vtScanFrame;
};

struct TObjRecoResult {
  string strObjTag;             // class_id of the best match
  Mat4x4F tWorld2Cam;           // row-major 4x4, last row 0 0 0 1
};

struct AdvancedParam {
  bool bEnablePoseBinFrameMatching;
  bool bEnablePreprocessing;
};

struct TTrainParam {
  int nType;
  int nMethod;
  bool bPreprocessing;
  float img_physical_width;
};
#endif  // __COMMON_H__

#ifndef __OBJ_RECO_TEMP__       // the reference's obj_reco_temp.h guard
class CObjRecoCAD {
 public:
  enum EObjRecoType { EObjReco_FEATURE, EObjReco_LmICP, EObjReco_BB8, EObjReco_PoseNet };
  virtual ~CObjRecoCAD() {}
  static string GetVersion();
  static CObjRecoCAD *Create(EObjRecoType eType = EObjReco_LmICP);   // nullptr for unsupported types
  static void Destroy(CObjRecoCAD *pHandle);
  virtual int Train(const string &strDataBase, const TScanPackage &tScanPackage, const TTrainParam &tObjTrainParam) = 0;
  virtual int AddObj(const string pObjModel) = 0;
  virtual int ClearObj() = 0;
  virtual int SetROI(const TImageU &tROI) = 0;
  virtual int Recognition(const TImageU &tRGB, const TImageU16 &tDepth, const TCamIntrinsicParam &tCamIntrinsic,
                          vector<TObjRecoResult> &vtResult) = 0;
  virtual int SetAdvancedParam(const AdvancedParam &advancedParam) = 0;
  virtual int GetAdvancedParam(const string &strKey, void *pvValue) = 0;
};
#endif

// ---- template bank files (linemod/linemod_if.h readLinemod / writeLinemod, without cv::Ptr) ----
namespace fealess {

struct Feature { int x, y, label; };
struct Template {
  int width, height, offset_x, offset_y, pyramid_level;
  std::vector<Feature> features;
};
struct ObjectClass {
  std::string class_id;
  std::vector<std::vector<Template> > template_pyramids;   // [template_id][l*M + m]
  std::vector<std::vector<float> > poses;                   // [template_id][13]
};
struct DetectorFile {            // what linemod_templates.yml holds (linemod.cpp:1681-1794)
  int pyramid_levels = 0;
  std::vector<int> T;
  std::vector<std::string> modalities;                       // "ColorGradient", "DepthNormal"
  std::vector<ObjectClass> classes;
};

// readLinemod (linemod_if.cpp:36-47): OpenCV FileStorage YAML 1.0 subset; false + err on failure
bool ReadLinemod(const std::string &filename, DetectorFile &out, std::string *err);
// writeLinemod (linemod_if.cpp:49-63)
bool WriteLinemod(const DetectorFile &det, const std::string &filename);
// packed binary cache of the same content (<yml>.flbank, keyed by the YAML's size and mtime); ReadLinemodCached =
// readLinemod that uses / refreshes it
bool WriteBankCache(const DetectorFile &det, const std::string &filename, unsigned long long yml_size, long long yml_mtime);
bool ReadBankCache(const std::string &filename, DetectorFile &out, unsigned long long yml_size, long long yml_mtime);
bool ReadLinemodCached(const std::string &filename, DetectorFile &out, std::string *err, bool *from_cache);
// imread(path, -1) for the 16-bit single-channel depth PNGs (obj_reco_lmicp.cpp:157)
bool ReadPng16(const std::string &filename, std::vector<unsigned short> &pixels, int &w, int &h, std::string *err);

}  // namespace fealess

// Extensions of the MI355X build (not in the reference): batch entry point on the same object.
class CObjRecoLmICPHip;
// out[i] = what Recognition() puts into vtResult for frame i; frame_rc (optional) = what it would return for frame i.
// Returns the first non-zero frame code: a frame that fails does not take the others' results with it.
int CadRecoRecognitionBatch(CObjRecoCAD *handle, int n_frames, const TImageU *rgb, const TImageU16 *depth,
                            const TCamIntrinsicParam &K, std::vector<std::vector<TObjRecoResult> > &out,
                            std::vector<int> *frame_rc = nullptr);

// Opt-in, not in the reference (its Recognition() only ever looks at matches[0]): refine the first k matches of every
// frame and return the nonMaximumSuppression (ICP/NMS.cpp:6-40, th_obj_dist = nms_dist_mm) winners, best first, in
// vtResult.  k = 1 restores the reference behaviour.
int CadRecoSetMultiHypothesis(CObjRecoCAD *handle, int k, float nms_dist_mm);

#endif  // FEALESS_CADRECO_H
