// A C++ caller holding a CObjRecoCAD* (what test/linemod_recon.cpp:35-80 of the reference does): Create, AddObj(dir),
// Recognition on a raw BGR8 + depth16 frame read from two files, pose printed with full precision, Destroy.
// Compiled on the GPU box against fealess_cadreco.h (tests/test_dropin_cpu.py proves in the container that the
// reference's own headers give the same layouts and link the same way).
#include "fealess_cadreco.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
static std::vector<unsigned char> slurp(const char *p)
{
  std::vector<unsigned char> b;
  FILE *f = fopen(p, "rb");
  if (!f) return b;
  fseek(f, 0, SEEK_END);
  b.resize((size_t)ftell(f));
  fseek(f, 0, SEEK_SET);
  if (fread(b.data(), 1, b.size(), f) != b.size()) b.clear();
  fclose(f);
  return b;
}
int main(int argc, char **argv)
{
  if (argc < 10) return 2;
  const int w = atoi(argv[4]), h = atoi(argv[5]);
  std::vector<unsigned char> bgr = slurp(argv[2]), dep = slurp(argv[3]);
  if (bgr.size() != (size_t)w * h * 3 || dep.size() != (size_t)w * h * 2) return 3;
  CObjRecoCAD *o = CObjRecoCAD::Create(CObjRecoCAD::EObjReco_LmICP);
  if (!o) return 4;
  printf("addobj %d\n", o->AddObj(argv[1]));
  TImageU rgb = {1.0, bgr.data(), w, h};
  TImageU16 d16 = {1.0, (unsigned short *)dep.data(), w, h};
  TCamIntrinsicParam K;
  K.nWidth = w; K.nHeight = h; K.dFx = atof(argv[6]); K.dFy = atof(argv[7]); K.dCx = atof(argv[8]); K.dCy = atof(argv[9]);
  vector<TObjRecoResult> out;
  const int rc = o->Recognition(rgb, d16, K, out);
  printf("recognition %d %zu\n", rc, out.size());
  if (!out.empty()) {
    printf("tag %s\npose", out[0].strObjTag.c_str());
    for (int i = 0; i < 16; ++i) printf(" %.9g", out[0].tWorld2Cam[i]);
    printf("\n");
  }
  CObjRecoCAD::Destroy(o);
  return 0;
}
