// A C++ caller of the multi-GPU host (include/fealess_mg.h): what a CadReco process adds around its per-rank detector.
// Compiled and linked against libfealess_mg.so by tests/test_dropin_cpu.py; without a GPU it only exercises the argument checks.
#include <cstdio>
#include <cstring>
#include <vector>
#include "fealess_mg.h"

int main(int argc, char **argv)
{
  (void)argc; (void)argv;
  char id[FL_MG_ID_BYTES];
  memset(id, 0, sizeof id);
  fl_mg *mg = nullptr;
  // no detector: rejected before RCCL is touched
  const int rc_null = fl_mg_create(nullptr, id, 8, 3, 6000, 2000, 64, &mg);
  printf("create_null_detector %d %d\n", rc_null, mg == nullptr ? 1 : 0);
  printf("unique_id_short_buffer %d\n", fl_mg_unique_id(id, 16));
  std::vector<fl_mg_result> res(4);
  fl_intrinsics K = {640, 480, 608.0, 608.0, 320.0, 240.0};
  fl_recognition_params p = {75.0f, 10, 0.5f, 0.01f, FL_ICP_PARITY};
  const uint8_t *bgr[4] = {nullptr, nullptr, nullptr, nullptr};
  const uint16_t *depth[4] = {nullptr, nullptr, nullptr, nullptr};
  printf("recognize_null_group %d\n", fl_mg_recognize_batch(nullptr, 4, bgr, depth, FL_MEM_HOST, &K, &p, res.data()));
  printf("last_error_null \"%s\"\n", fl_mg_last_error(nullptr));
  printf("sizeof_result %zu\n", sizeof(fl_mg_result));
  fl_mg_destroy(nullptr);
  return 0;
}
