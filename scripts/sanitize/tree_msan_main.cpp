// MemorySanitizer driver for the TREE pipeline's source (csrc/sg_tree.h) on the host.  TEST INFRASTRUCTURE.
//
// ASan sees an access outside an array, not a READ OF A VALUE NOBODY WROTE -- and that is what differs between the host and the GPU:
// a local variable left uninitialised is whatever the register held (it changes with every build of the kernel), a word of the env's
// LDS block is whatever the previous workgroup on that CU left there.  This program replays oracle states (scripts/sanitize/run_tree_msan.sh
// dumps them: model blob + per-substep qpos / qvel / act / warmstart / ctrl) through the emulation's entry points with the LDS-class
// arrays marked uninitialised before every launch; MSan reports the first branch, address or division that depends on such a value.
#include <sanitizer/msan_interface.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define SGT_EMU_SEPARATE
#define SGT_EMU_MSAN
#include "../../tests/emu/sg_tree_emu.cpp"

static std::vector<char> slurp(const char* p) {
  FILE* f = fopen(p, "rb");
  if (!f) { perror(p); exit(2); }
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  std::vector<char> b(n);
  if (fread(b.data(), 1, n, f) != (size_t)n) exit(2);
  fclose(f);
  return b;
}

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s model.blob states.bin\n", argv[0]); return 2; }
  std::vector<char> blob = slurp(argv[1]), st = slurp(argv[2]);
  char err[256];
  TreeEmu* E = temu_new(blob.data(), blob.size(), err, sizeof err);
  if (!E) { fprintf(stderr, "%s\n", err); return 1; }
  const char* p = st.data();
  double k; int nj, nt;
  memcpy(&k, p, 8); memcpy(&nj, p + 8, 4); memcpy(&nt, p + 12, 4); p += 16;
  temu_set_stiffness(E, k, (const int*)p, nj, (const int*)(p + 4 * nj), nt); p += 4 * (nj + nt);
  int nrec; memcpy(&nrec, p, 4); p += 4;
  const int nq = temu_nq(E), nv = temu_nv(E), nu = temu_nu(E);
  temu_run(E, 1, 1);
  for (int r = 0; r < nrec; r++) {
    const double* d = (const double*)p; p += 8 * (size_t)(nq + 3 * nv - nv + 2 * nu);   // qpos, qvel, act, warm, ctrl
    memcpy(temu_qpos(E), d, 8 * nq); d += nq;
    memcpy(temu_qvel(E), d, 8 * nv); d += nv;
    memcpy(temu_act(E), d, 8 * nu); d += nu;
    memcpy(temu_warm(E), d, 8 * nv); d += nv;
    memcpy(temu_ctrl(E), d, 8 * nu);
    temu_run(E, 0, 1);
    if (r % 7 == 6) printf("step %d: ncon %d nefc %d iters %d flags %d\n", r / 7, temu_ncon(E), temu_nefc(E), temu_iters(E), temu_flags(E));
  }
  temu_free(E);
  return 0;
}
