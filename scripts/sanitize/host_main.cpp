#include <cstdio>
#include <string>
#include "../../soft-grip_amd/csrc/sg_mjcf.h"
#include "../../soft-grip_amd/csrc/sg_plan.h"
int main(int argc, char** argv) {
  for (int i = 1; i < argc; i++) {
    for (int nb = 0; nb < 2; nb++) {
      std::string blob, err;
      bool ok = sg_mjcf_compile_file(argv[i], nb, false, &blob, &err);
      if (!ok) { printf("%s: compile error: %s\n", argv[i], err.c_str()); continue; }
      SgPlan P;
      std::string e2;
      bool pk = sg_plan_build(blob.data(), blob.size(), &P, &e2);
      printf("%s nb=%d: blob %zu bytes, plan %s %s\n", argv[i], nb, blob.size(), pk ? "ok" : "refused:", pk ? "" : e2.c_str());
    }
  }
  return 0;
}
