// A caller compiled ONLY against the reference's own headers (-I<reference>/CadReco: obj_reco_temp.h, lotus_common.h),
// linked against libcadreco_hip.so: prints the layout table as the reference defines the types, then drives the library
// through a CObjRecoCAD* exactly like test/linemod_recon.cpp:35-80 does -- Create, AddObj, Recognition, Destroy.
// Without a GPU the calls must return the documented error codes (no crash, no CPU path).
#include "obj_reco_temp.h"
#include <cstdio>
#include <cstring>
#include "layout_table.inc"
int main(int argc, char **argv)
{
  for (const auto &r : LAYOUT_TABLE) printf("%s %lld\n", r.name, r.value);
  printf("--calls\n");
  printf("version %s\n", CObjRecoCAD::GetVersion().substr(0, 30).c_str());
  printf("create_unsupported %d\n", CObjRecoCAD::Create(CObjRecoCAD::EObjReco_BB8) == nullptr ? 1 : 0);
  CObjRecoCAD *h = CObjRecoCAD::Create();                 // default EObjReco_LmICP
  printf("create %d\n", h != nullptr ? 1 : 0);
  if (!h) return 1;
  printf("addobj_missing %d\n", h->AddObj(argc > 1 ? argv[1] : "/nonexistent/dir"));
  printf("clearobj %d\n", h->ClearObj());
  AdvancedParam ap = {false, false};
  printf("setadvanced %d\n", h->SetAdvancedParam(ap));
  static unsigned char bgr[640 * 480 * 3];
  static unsigned short depth[640 * 480];
  TImageU rgb = {0.0, bgr, 640, 480};
  TImageU16 dep = {0.0, depth, 640, 480};
  TCamIntrinsicParam K;
  K.nWidth = 640; K.nHeight = 480; K.dFx = 608; K.dFy = 608; K.dCx = 320; K.dCy = 240;
  vector<TObjRecoResult> out;
  const int rc = h->Recognition(rgb, dep, K, out);
  printf("recognition_no_object %d %zu\n", rc, out.size());
  TImageU bad = {-1.0, bgr, 640, 480};                    // negative timestamp: CheckTImage fails (obj_reco_lmicp.cpp:32-36)
  printf("recognition_bad_image %d\n", h->Recognition(bad, dep, K, out));
  CObjRecoCAD::Destroy(h);
  printf("destroyed 1\n");
  return 0;
}
