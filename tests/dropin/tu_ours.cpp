// the layout table as fealess_amd/cadreco/fealess_cadreco.h defines the types
#include "fealess_cadreco.h"
#include <cstdio>
#include "layout_table.inc"
int main()
{
  for (const auto &r : LAYOUT_TABLE) printf("%s %lld\n", r.name, r.value);
  return 0;
}
