// sg_api.hip -- C ABI (include/softgrip.h) over the gfx950 kernels.  No CPU fallback.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "sg_mjcf.h"
#include "sg_tree.h"
#include "sg_work.h"
#ifdef SG_LEGACY_PIPELINES
#include "sg_kernels_args.h"
#endif

#ifdef SG_LEGACY_PIPELINES
#define SG_LEGACY_ON 1
#else
#define SG_LEGACY_ON 0   // (the split pipeline's contact records, 126 MB at 4096 envs, are then not allocated)
#endif

namespace {
thread_local std::string g_err;
int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIPCHK(x)                                                                              \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess) return fail(SG_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); \
  } while (0)

__global__ void sg_fill_rows_kernel(double* dst, const double* row, int n, int w) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n * w) dst[i] = row[i % w];
}
#define SG_CTRL_BYVAL 8
struct SgCtrlRow { double v[SG_CTRL_BYVAL]; };
__global__ void sg_fill_rows_val_kernel(double* dst, SgCtrlRow row, int n, int w) {  // the row travels in the kernel arguments: no host sync
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n * w) dst[i] = row.v[i % w];
}
__global__ void sg_masked_copy_kernel(const unsigned char* mask, int n, const int* s0, int* d0, const int* s1, int* d1, const int* s2, int* d2,
                                      const int* s3, int* d3, const int* s4, int* d4) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && (!mask || mask[i])) { d0[i] = s0[i]; d1[i] = s1[i]; d2[i] = s2[i]; d3[i] = s3[i]; d4[i] = s4[i]; }
}
}  // namespace

struct sg_model {
  SgPlan plan;      // the fast kernels' plan (has_fast), else a copy of tplan: header, elements and sizes serve every entry point
  int rounds;       // ceil(nelem / 64)
  bool has_fast;    // the model is in the two-finger class of sg_plan.h
  bool has_tree;    // the tree pipeline (sg_tree.h) runs it
  SgPlan tplan;     // the tree pipeline's plan: same elements / equalities / statics, chains in `tree`, flat box references
  SgTreeDev tree;
};

struct sg_batch {
  const sg_model* m;
  int n, device;
  SgPlanHeader* dH;
  double *delem, *qpos, *qvel, *warm, *act, *ctrl, *kenv, *ctrl_row;
  SgGenPair* dgpairs;  // SgPlan::gpairs on the device (the general contact path's candidate pairs)
  int* dnbtab;       // SgPlan::nbtab on the device (neighbour-row models)
  SgEqSlot* dsched;  // SgPlan::sched + one spare round of idle slots
  unsigned* dtab;    // the same schedule as the solver's LDS table words
  int* dcpos;        // per element: where its equality block's step factors sit in a solver wavefront's stream (SgWork::cst)
  int *kmask_jnt, *kmask_ten, *flags, *touch, *ncon, *nefc, *iters;
  std::vector<int> kmask_jnt_host, kmask_ten_host;   // what the device masks hold (sg_set_stiffness copies them only when they change)
  int epw_override;  // sg_set_solver_envs_per_wavefront: 0 = automatic
  int pipeline;  // 0 fused (one kernel per call), 1 split (chain / phase / pgs kernel chain), 2 split with the row-parallel PGS kernel, 3 tree
  // tree pipeline (allocated when first selected)
  SgPlanHeader* dTH;
  SgTreeDev* dT;
  double *dtelem, *tcws;
  SgGenPair* dtpairs;
  SgEqSlot* dtsched;  // neighbour-row models: the tree plan's block schedule and neighbour tables
  int* dtnbtab;
  int* touch_words;   // [n][2]
  bool tree_ready;    // every table and the work space of the tree pipeline allocated and filled (tree_alloc)
  bool tree_attr_set;
  SgWork w;
  std::vector<void*> wbufs;
  bool lds_attr_set;  // hipFuncSetAttribute(MaxDynamicSharedMemorySize) done on this batch's device
  // profiling
  bool prof;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
  double prof_ms;
  long long prof_n;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pgs;  // around every solver-kernel launch (the dominant kernel)
  std::vector<hipEvent_t> ev_pool;  // events handed back by sg_profile_read*: a profiled call creates none once the pool is warm
  double prof_pgs_ms;
  long long prof_pgs_n;
};

static int begin_event_pair(sg_batch* b, std::vector<std::pair<hipEvent_t, hipEvent_t>>& list, hipStream_t s);

// device tables and work space of the tree pipeline (sg_tree.h), allocated when the pipeline is first selected
static void tree_free(sg_batch* b) {
  void** tptrs[] = {(void**)&b->dTH, (void**)&b->dT, (void**)&b->dtelem, (void**)&b->tcws, (void**)&b->dtpairs, (void**)&b->touch_words, (void**)&b->dtsched,
                    (void**)&b->dtnbtab};
  for (void** p : tptrs) {
    if (*p) (void)hipFree(*p);
    *p = nullptr;
  }
  b->tree_ready = false;
}
static int tree_alloc_all(sg_batch* b);
// all or nothing: a failed allocation (the work space is ~0.5 MB per env for the four-finger scene) leaves no half-built pipeline behind
// that a second sg_set_pipeline call would take for a complete one
static int tree_alloc(sg_batch* b) {
  if (b->tree_ready) return SG_OK;
  const int rc = tree_alloc_all(b);
  if (rc != SG_OK) tree_free(b);
  else b->tree_ready = true;
  return rc;
}
static int tree_alloc_all(sg_batch* b) {
  const sg_model* m = b->m;
  if (!m->has_tree) return fail(SG_ERR_MODEL, "the tree pipeline does not run this model");
  const size_t n = b->n;
  const long long cwd = sgt::cws_doubles(m->tree, m->tplan.h.nelem, m->tplan.h.has_free, m->tplan.h.nnb);
#define TALLOC(p, bytes)                                                                  \
  do {                                                                                    \
    hipError_t e_ = hipMalloc((void**)&(p), (bytes));                                     \
    if (e_ != hipSuccess) return fail(SG_ERR_NOMEM, std::string("hipMalloc (tree pipeline): ") + hipGetErrorString(e_)); \
  } while (0)
  TALLOC(b->dTH, sizeof(SgPlanHeader));
  TALLOC(b->dT, sizeof(SgTreeDev));
  TALLOC(b->dtelem, sizeof(double) * m->tplan.elem.size());
  TALLOC(b->dtpairs, sizeof(SgGenPair) * (m->tplan.gpairs.size() + 1));
  TALLOC(b->tcws, sizeof(double) * n * (size_t)cwd);
  TALLOC(b->touch_words, sizeof(int) * 2 * n);
  TALLOC(b->dtsched, sizeof(SgEqSlot) * (m->tplan.sched.size() + 1));
  TALLOC(b->dtnbtab, sizeof(int) * (m->tplan.nbtab.size() + 1));
#undef TALLOC
  HIPCHK(hipMemcpy(b->dtsched, m->tplan.sched.data(), sizeof(SgEqSlot) * m->tplan.sched.size(), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(b->dtnbtab, m->tplan.nbtab.data(), sizeof(int) * m->tplan.nbtab.size(), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(b->dTH, &m->tplan.h, sizeof(SgPlanHeader), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(b->dT, &m->tree, sizeof(SgTreeDev), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(b->dtelem, m->tplan.elem.data(), sizeof(double) * m->tplan.elem.size(), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(b->dtpairs, m->tplan.gpairs.data(), sizeof(SgGenPair) * m->tplan.gpairs.size(), hipMemcpyHostToDevice));
  HIPCHK(hipMemset(b->tcws, 0, sizeof(double) * n * (size_t)cwd));
  HIPCHK(hipMemset(b->touch_words, 0, sizeof(int) * 2 * n));
  if (!b->w.secprof) {   // (a model outside the two-finger class has no split-pipeline work space)
    void* p = nullptr;
    if (hipMalloc(&p, sizeof(unsigned long long) * 48) != hipSuccess) return fail(SG_ERR_NOMEM, "hipMalloc (tree pipeline)");
    b->wbufs.push_back(p);
    b->w.secprof = (unsigned long long*)p;
    HIPCHK(hipMemset(p, 0, sizeof(unsigned long long) * 48));
  }
  return SG_OK;
}

// tree pipeline: one launch per call, one env per wavefront (sg_tree.hip)
static int launch_tree(sg_batch* b, int mode, const uint8_t* mask, int nsub, double* sens, long long stride, int32_t* flags, int32_t* touch,
                       hipStream_t s) {
  const sg_model* m = b->m;
  sgt::TreeArgs a;
  a.H = b->dTH; a.T = b->dT; a.elem = b->dtelem; a.gpairs = b->dtpairs; a.sched = b->dtsched; a.nbtab = b->dtnbtab;
  a.qpos = b->qpos; a.qvel = b->qvel; a.warm = b->warm; a.act = b->act; a.ctrl = b->ctrl;
  a.kenv = b->kenv; a.kmask_jnt = b->kmask_jnt; a.kmask_ten = b->kmask_ten;
  a.mask = mask; a.sens = sens; a.sens_stride = stride > 0 ? stride : m->tplan.h.nsensordata;
  a.flags = flags ? flags : b->flags; a.touch = touch ? touch : b->touch; a.touch_words = b->touch_words;
  a.ncon = b->ncon; a.nefc = b->nefc; a.iters = b->iters;
  a.cws = b->tcws; a.cws_stride = sgt::cws_doubles(m->tree, m->tplan.h.nelem, m->tplan.h.has_free, m->tplan.h.nnb);
  a.nenv = b->n; a.nsub = nsub; a.mode = mode; a.secprof = b->w.secprof;
  const size_t lds = sgt::lds_bytes(m->tree, m->tplan.h.nelem, m->tplan.h.has_free, m->tplan.h.nnb);
  if (!b->tree_attr_set) {
    HIPCHK(sg_tree_prepare());
    b->tree_attr_set = true;
  }
  if (b->prof)
    if (int rc = begin_event_pair(b, b->ev, s)) return rc;
  HIPCHK(sg_launch_tree(a, m->tree.CS, lds, s));
  if (b->prof) HIPCHK(hipEventRecord(b->ev.back().second, s));
  return SG_OK;
}

extern "C" {

const char* sg_last_error(void) { return g_err.c_str(); }
const char* sg_version(void) { return "softgrip-mi355x 0.1 (gfx950)"; }

int sg_model_create(const void* blob, size_t nbytes, sg_model** out) {
  if (!blob || !out) return fail(SG_ERR_INVALID, "sg_model_create: null argument");
  sg_model* m = new sg_model();
  std::string err, terr;
  m->has_fast = sg_plan_build(blob, nbytes, &m->plan, &err);
  m->has_tree = sg_tree_plan_build(blob, nbytes, &m->tplan, &m->tree, &terr);
  if (m->has_tree && sgt::lds_bytes(m->tree, m->tplan.h.nelem, m->tplan.h.has_free, m->tplan.h.nnb) > 160 * 1024) {
    m->has_tree = false;
    terr = "the env's state does not fit the 160 KB of LDS";
  }
  if (!m->has_fast && !m->has_tree) {
    delete m;
    return fail(SG_ERR_MODEL, "sg_model_create: " + err + " (two-finger kernels); " + terr + " (tree pipeline)");
  }
  if (!m->has_fast) m->plan = m->tplan;
  m->rounds = (m->plan.h.nelem + 63) / 64;
  if (m->rounds > 4) {
    delete m;
    return fail(SG_ERR_MODEL, "sg_model_create: more than 256 composite elements");
  }
  if (m->has_fast && m->plan.h.nnb > 0) {  // the rows PGS kernel keeps every equality row of 8 envs in LDS (launch_split)
    if (sizeof(double) * SG_ROWS_LDS_NB(4, m->plan.h.nelem, m->plan.h.eq_rounds, 1) > 160 * 1024 || m->plan.h.nelem > 254) {  // (11-bit slider offsets in the table words)
      delete m;
      return fail(SG_ERR_MODEL, "sg_model_create: too many neighbour equality rows for the PGS kernel's LDS");
    }
  }
  *out = m;
  return SG_OK;
}
void sg_model_destroy(sg_model* m) { delete m; }

int sg_mjcf_compile(const char* xml_path, int flags, void** blob, size_t* nbytes) {
  if (!xml_path || !blob || !nbytes) return fail(SG_ERR_INVALID, "sg_mjcf_compile: null argument");
  std::string out, err;
  if (!sg_mjcf_compile_file(xml_path, !(flags & SG_COMPILE_NO_NEIGHBORS), (flags & SG_COMPILE_IMPLICIT_TENDON_DAMPER) != 0, &out, &err))
    return fail(SG_ERR_MODEL, "sg_mjcf_compile: " + err);
  void* p = malloc(out.size());
  if (!p) return fail(SG_ERR_INVALID, "sg_mjcf_compile: out of memory");
  memcpy(p, out.data(), out.size());
  *blob = p; *nbytes = out.size();
  return SG_OK;
}
void sg_blob_free(void* blob) { free(blob); }

int sg_model_compile(const char* xml_path, int flags, sg_model** out) {
  void* blob = nullptr;
  size_t nbytes = 0;
  int rc = sg_mjcf_compile(xml_path, flags, &blob, &nbytes);
  if (rc != SG_OK) return rc;
  rc = sg_model_create(blob, nbytes, out);
  free(blob);
  return rc;
}
int sg_model_nq(const sg_model* m) { return m->plan.h.nq; }
int sg_model_nv(const sg_model* m) { return m->plan.h.nv; }
int sg_model_njnt(const sg_model* m) { return m->plan.h.njnt; }
int sg_model_nu(const sg_model* m) { return m->plan.h.nu; }
int sg_model_nsensordata(const sg_model* m) { return m->plan.h.nsensordata; }
int sg_model_ntendon(const sg_model* m) { return m->plan.h.ntendon; }
int sg_model_nelem(const sg_model* m) { return m->plan.h.nelem; }

void sg_batch_destroy(sg_batch* b) {
  if (!b) return;
  (void)hipSetDevice(b->device);
  tree_free(b);
  void* ptrs[] = {b->dtab, b->dcpos, b->dgpairs, b->dnbtab, b->dsched, b->dH, b->delem, b->qpos, b->qvel, b->warm, b->act, b->ctrl, b->kenv, b->ctrl_row, b->kmask_jnt, b->kmask_ten,
                  b->flags, b->touch, b->ncon, b->nefc, b->iters};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  for (void* p : b->wbufs)
    if (p) (void)hipFree(p);
  for (auto& e : b->ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  for (auto& e : b->ev_pgs) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  for (auto& e : b->ev_pool) (void)hipEventDestroy(e);
  delete b;
}

int sg_batch_create(const sg_model* m, int n_envs, int device, sg_batch** out) {
  if (!m || !out || n_envs <= 0) return fail(SG_ERR_INVALID, "sg_batch_create: bad argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(SG_ERR_NO_DEVICE, "sg_batch_create: no HIP device (there is no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(SG_ERR_NO_DEVICE, "sg_batch_create: device index out of range");
  HIPCHK(hipSetDevice(device));
  sg_batch* b = new sg_batch();
  b->m = m; b->n = n_envs; b->device = device; b->lds_attr_set = false; b->prof = false; b->prof_ms = 0; b->prof_n = 0; b->prof_pgs_ms = 0; b->prof_pgs_n = 0;
  b->dnbtab = nullptr; b->dsched = nullptr; b->dgpairs = nullptr; b->dtab = nullptr; b->dcpos = nullptr; b->epw_override = 0;
  b->dTH = nullptr; b->dT = nullptr; b->dtelem = b->tcws = nullptr; b->dtpairs = nullptr; b->touch_words = nullptr; b->tree_ready = false; b->tree_attr_set = false; b->dtsched = nullptr; b->dtnbtab = nullptr;
  b->dH = nullptr; b->delem = b->qpos = b->qvel = b->warm = b->act = b->ctrl = b->kenv = b->ctrl_row = nullptr;
  b->kmask_jnt = b->kmask_ten = b->flags = b->touch = b->ncon = b->nefc = b->iters = nullptr;
  const SgPlanHeader& H = m->plan.h;
  const size_t n = n_envs, nv = H.nv, nq = H.nq, njnt = H.njnt, nu = H.nu > 0 ? H.nu : 1, nt = H.ntendon;   // nq = nv = njnt unless the model has a free joint
#define ALLOC(p, bytes)                                   \
  do {                                                    \
    hipError_t e_ = hipMalloc((void**)&(p), (bytes));     \
    if (e_ != hipSuccess) {                               \
      sg_batch_destroy(b);                                \
      return fail(SG_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e_)); \
    }                                                     \
  } while (0)
  ALLOC(b->dH, sizeof(SgPlanHeader));
  ALLOC(b->delem, sizeof(double) * m->plan.elem.size());
  ALLOC(b->qpos, sizeof(double) * n * nq); ALLOC(b->qvel, sizeof(double) * n * nv); ALLOC(b->warm, sizeof(double) * n * nv);
  ALLOC(b->act, sizeof(double) * n * nu); ALLOC(b->ctrl, sizeof(double) * n * nu); ALLOC(b->kenv, sizeof(double) * n);
  ALLOC(b->ctrl_row, sizeof(double) * nu);
  ALLOC(b->kmask_jnt, sizeof(int) * njnt); ALLOC(b->kmask_ten, sizeof(int) * nt);
  ALLOC(b->flags, sizeof(int) * n); ALLOC(b->touch, sizeof(int) * n); ALLOC(b->ncon, sizeof(int) * n); ALLOC(b->nefc, sizeof(int) * n);
  ALLOC(b->iters, sizeof(int) * n);
  memset(&b->w, 0, sizeof b->w);
  b->pipeline = 3;
  if (m->has_fast) {  // workspace of the split pipeline
    const size_t S = 2 * n, N = H.nelem;
    auto walloc = [&](void** p, size_t bytes) -> bool {
      if (hipMalloc(p, bytes) != hipSuccess) return false;
      b->wbufs.push_back(*p);
      return hipMemset(*p, 0, bytes) == hipSuccess;
    };
    bool ok = walloc((void**)&b->w.secprof, sizeof(unsigned long long) * 48) && walloc((void**)&b->w.crec, sizeof(double) * (SG_LEGACY_ON ? SG_CAP * ((n + SG_EPW - 1) / SG_EPW + 1) * SG_RF * SG_SPW : 2)) && walloc((void**)&b->w.ns, sizeof(int) * S) &&
              walloc((void**)&b->w.crow, sizeof(double) * (SG_CAP + 2) * ((n + 7) / 8 + 2) * SG_RK * 64) &&
              walloc((void**)&b->w.cdummy, sizeof(double) * ((n + 3) / 4) * SG_RK * 64) &&
              walloc((void**)&b->w.envh, sizeof(double) * 4 * n) && walloc((void**)&b->w.shared, sizeof(int) * n) &&
              walloc((void**)&b->w.pending, sizeof(int) * n) && walloc((void**)&b->w.status, sizeof(int) * n) &&
              walloc((void**)&b->w.iters, sizeof(int) * n) && walloc((void**)&b->w.ncon, sizeof(int) * n) &&
              walloc((void**)&b->w.nefc, sizeof(int) * n) && walloc((void**)&b->w.touch, sizeof(int) * n) &&
              walloc((void**)&b->w.sMinv, sizeof(double) * 16 * S) && walloc((void**)&b->w.saF, sizeof(double) * 4 * S) &&
              walloc((void**)&b->w.lim_active, sizeof(int) * S) && walloc((void**)&b->w.lim, sizeof(double) * 4 * SG_MAXLIM * S) &&
              walloc((void**)&b->w.as, sizeof(double) * n * N) && walloc((void**)&b->w.eqf, sizeof(double) * n * N) &&
              walloc((void**)&b->w.eqb, sizeof(double) * n * N) && walloc((void**)&b->w.eqR, sizeof(double) * n * N) &&
              walloc((void**)&b->w.asme, sizeof(double) * n * N) && walloc((void**)&b->w.fsm, sizeof(double) * n * N) &&
              walloc((void**)&b->w.chh, sizeof(double) * n * 2 * SG_CHW) &&
              walloc((void**)&b->w.nbf, sizeof(double) * n * (3 * N + 1)) && walloc((void**)&b->w.nbb, sizeof(double) * n * (3 * N + 1)) &&
              walloc((void**)&b->w.nbR, sizeof(double) * n * (3 * N + 1)) &&
              walloc((void**)&b->w.cst, sizeof(double) * ((H.nnb > 0 && SG_ROWS_NB_MODE(H.nelem, H.eq_rounds) == 2) ? SG_CST_INDEX((n + 3) / 4, 0, 0, H.eq_rounds + 8) : 2)) &&
              walloc((void**)&b->w.gcon, sizeof(double) * n * SG_GEN_MAXCON * SG_GEN_W) && walloc((void**)&b->w.gen, sizeof(int) * n) &&
              walloc((void**)&b->w.gen_count, sizeof(int) * 4) && walloc((void**)&b->w.gen_list, sizeof(int) * n) &&
              walloc((void**)&b->w.gpairs16, SG_PHASE_SLIM(m->rounds) ? sizeof(unsigned short) * n * SG_PAIRS_CAP(4) : 16) &&
              walloc((void**)&b->w.gcval, SG_PHASE_SLIM(m->rounds) ? sizeof(double) * n * SG_MAXCH * 64 : 16);
    if (!ok) { sg_batch_destroy(b); return fail(SG_ERR_NOMEM, "hipMalloc (split-pipeline workspace)"); }
    const char* pm = getenv("SG_PIPELINE");
    b->pipeline = (pm && strcmp(pm, "tree") == 0 && m->has_tree) ? 3 : 2;
#ifdef SG_LEGACY_PIPELINES
    if (pm && strcmp(pm, "fused") == 0) b->pipeline = 0;
    if (pm && strcmp(pm, "split") == 0) b->pipeline = 1;
#endif
    if (H.nnb > 0 && b->pipeline != 3) b->pipeline = 2;  // neighbour equality rows: the rows pipeline (or the tree pipeline when asked for)
  }
  if (m->has_fast && H.nnb > 0) {
    std::vector<SgEqSlot> sch = m->plan.sched;
    SgEqSlot idle;
    idle.e = idle.p[0] = idle.p[1] = idle.p[2] = H.nelem;
    for (int g = 0; g < 8 * SG_EQ_SLOTS; g++) sch.push_back(idle);  // the kernel fetches slots a few rounds ahead
    ALLOC(b->dnbtab, sizeof(int) * m->plan.nbtab.size());
    ALLOC(b->dsched, sizeof(SgEqSlot) * sch.size());
    HIPCHK(hipMemcpy(b->dnbtab, m->plan.nbtab.data(), sizeof(int) * m->plan.nbtab.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b->dsched, sch.data(), sizeof(SgEqSlot) * sch.size(), hipMemcpyHostToDevice));
    // the solver's table words (sg_pgs_rows_kernel): lane 2 b + h of a 16-lane group holds, for the block e in slot b of the round,
    // x | y << 11 | (2 e + h) << 22: the byte offsets of two slider words, (x, y) = 8 (e, p0) for h = 0 and 8 (p1, p2) for h = 1, and the
    // lane's pair of row states in 16-byte units
    // (models that keep the step factors in LDS -- SG_ROWS_NB_MODE 1, the box scene -- get TWO words per lane and round instead:
    //  x | y << 16 and the byte offset of the lane's pair of 32-byte records)
    const bool two_words = SG_ROWS_NB_MODE(H.nelem, H.eq_rounds) == 1;
    std::vector<unsigned> tab((two_words ? 4 : 2) * sch.size());
    for (size_t i = 0; i < 2 * sch.size(); i++) {
      const SgEqSlot& sl = sch[i >> 1];
      const int h = (int)(i & 1);
      const unsigned x = h ? sl.p[1] : sl.e, y = h ? sl.p[2] : sl.p[0];
      if (two_words) { tab[2 * i] = (8u * x) | ((8u * y) << 16); tab[2 * i + 1] = 64u * (unsigned)sl.e + 32u * (unsigned)h; }
      else tab[i] = (8u * x) | ((8u * y) << 11) | ((2u * (unsigned)sl.e + (unsigned)h) << 22);
    }
    ALLOC(b->dtab, sizeof(unsigned) * tab.size());
    HIPCHK(hipMemcpy(b->dtab, tab.data(), sizeof(unsigned) * tab.size(), hipMemcpyHostToDevice));
    // where the phase kernel puts a block's four step factors: round r, slot g of the schedule = lanes 2 g, 2 g + 1 of the env's
    // 16-lane group, two doubles each (SG_CST_INDEX)
    std::vector<int> cpos(H.nelem, 0);
    for (size_t i = 0; i < m->plan.sched.size(); i++)
      if (m->plan.sched[i].e < H.nelem) cpos[m->plan.sched[i].e] = (int)(i / SG_EQ_SLOTS) * 128 + 4 * (int)(i % SG_EQ_SLOTS);
    ALLOC(b->dcpos, sizeof(int) * cpos.size());
    HIPCHK(hipMemcpy(b->dcpos, cpos.data(), sizeof(int) * cpos.size(), hipMemcpyHostToDevice));
  }
  ALLOC(b->dgpairs, sizeof(SgGenPair) * (m->plan.gpairs.size() + 1));
  HIPCHK(hipMemcpy(b->dgpairs, m->plan.gpairs.data(), sizeof(SgGenPair) * m->plan.gpairs.size(), hipMemcpyHostToDevice));
#undef ALLOC
  HIPCHK(hipMemcpy(b->dH, &H, sizeof H, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(b->delem, m->plan.elem.data(), sizeof(double) * m->plan.elem.size(), hipMemcpyHostToDevice));
  // state as after mj_resetData
  std::vector<double> q0(nq, 0.0);
  for (int c = 0; c < H.nchain; c++)
    for (int d = 0; d < H.chain[c].ndof; d++) q0[H.chain[c].dof0 + d] = H.chain[c].qpos0[d];
  if (!m->has_fast)
    for (int d = 0; d < m->tree.ND; d++) q0[m->tree.d_gid[d]] = m->tree.d_qpos0[d];
  for (int e = 0; e < H.nelem; e++) q0[H.elem_qpos0 + e] = m->plan.elem[(size_t)SGE_QPOS0 * H.nelem + e];
  if (H.has_free)
    for (int c = 0; c < 7; c++) q0[H.free_qadr + c] = H.free_q0[c];
  std::vector<double> qall(n * nq);
  for (size_t i = 0; i < n; i++) memcpy(&qall[i * nq], q0.data(), sizeof(double) * nq);
  HIPCHK(hipMemcpy(b->qpos, qall.data(), sizeof(double) * n * nq, hipMemcpyHostToDevice));
  HIPCHK(hipMemset(b->qvel, 0, sizeof(double) * n * nv)); HIPCHK(hipMemset(b->warm, 0, sizeof(double) * n * nv));
  HIPCHK(hipMemset(b->act, 0, sizeof(double) * n * nu)); HIPCHK(hipMemset(b->ctrl, 0, sizeof(double) * n * nu));
  HIPCHK(hipMemset(b->kenv, 0, sizeof(double) * n));
  HIPCHK(hipMemset(b->kmask_jnt, 0, sizeof(int) * njnt)); HIPCHK(hipMemset(b->kmask_ten, 0, sizeof(int) * nt));
  b->kmask_jnt_host.assign(njnt, 0); b->kmask_ten_host.assign(nt, 0);
  HIPCHK(hipMemset(b->flags, 0, sizeof(int) * n)); HIPCHK(hipMemset(b->touch, 0, sizeof(int) * n));
  HIPCHK(hipMemset(b->ncon, 0, sizeof(int) * n)); HIPCHK(hipMemset(b->nefc, 0, sizeof(int) * n)); HIPCHK(hipMemset(b->iters, 0, sizeof(int) * n));
  if (b->pipeline == 3)
    if (int rc = tree_alloc(b)) { sg_batch_destroy(b); return rc; }
  *out = b;
  return SG_OK;
}
int sg_batch_nenvs(const sg_batch* b) { return b->n; }
int sg_batch_device(const sg_batch* b) { return b->device; }

int sg_set_stiffness(sg_batch* b, const double* k, int k_on_host, const int* jnt_ids, int nj, const int* ten_ids, int nt, void* stream) {
  if (!b || !k || nj < 0 || nt < 0 || (nj && !jnt_ids) || (nt && !ten_ids)) return fail(SG_ERR_INVALID, "sg_set_stiffness: bad argument");
  HIPCHK(hipSetDevice(b->device));
  const SgPlanHeader& H = b->m->plan.h;
  std::vector<int> mj(H.njnt, 0), mt(H.ntendon, 0);
  for (int i = 0; i < nj; i++) {
    if (jnt_ids[i] < 0 || jnt_ids[i] >= H.njnt) return fail(SG_ERR_INVALID, "sg_set_stiffness: joint id out of range");
    mj[jnt_ids[i]] = 1;
  }
  for (int i = 0; i < nt; i++) {
    if (ten_ids[i] < 0 || ten_ids[i] >= H.ntendon) return fail(SG_ERR_INVALID, "sg_set_stiffness: tendon id out of range");
    mt[ten_ids[i]] = 1;
  }
  hipStream_t s = (hipStream_t)stream;
  // Host synchronisation (softgrip.h says so): the id masks are tiny and change once per scene, not once per episode -- they are
  // copied (synchronously: the host vectors' lifetime stays trivial) only when they differ from what the device holds; a HOST k is a
  // synchronous copy by nature (pageable memory).  With unchanged id sets and a DEVICE k the call enqueues one copy and returns.
  const bool masks_changed = mj != b->kmask_jnt_host || mt != b->kmask_ten_host;
  if (masks_changed || k_on_host) HIPCHK(hipStreamSynchronize(s));
  if (masks_changed) {
    HIPCHK(hipMemcpy(b->kmask_jnt, mj.data(), sizeof(int) * H.njnt, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b->kmask_ten, mt.data(), sizeof(int) * H.ntendon, hipMemcpyHostToDevice));
    b->kmask_jnt_host = mj;
    b->kmask_ten_host = mt;
  }
  if (k_on_host) HIPCHK(hipMemcpy(b->kenv, k, sizeof(double) * b->n, hipMemcpyHostToDevice));
  else HIPCHK(hipMemcpyAsync(b->kenv, k, sizeof(double) * b->n, hipMemcpyDeviceToDevice, s));
  return SG_OK;
}

int sg_set_ctrl(sg_batch* b, const double* ctrl, int broadcast, void* stream) {
  if (!b || !ctrl) return fail(SG_ERR_INVALID, "sg_set_ctrl: bad argument");
  HIPCHK(hipSetDevice(b->device));
  hipStream_t s = (hipStream_t)stream;
  const int nu = b->m->plan.h.nu;
  if (nu == 0) return SG_OK;
  if (broadcast && nu <= SG_CTRL_BYVAL) {
    SgCtrlRow row;
    for (int i = 0; i < SG_CTRL_BYVAL; i++) row.v[i] = i < nu ? ctrl[i] : 0.0;
    int total = b->n * nu;
    hipLaunchKernelGGL(sg_fill_rows_val_kernel, dim3((total + 255) / 256), dim3(256), 0, s, b->ctrl, row, b->n, nu);
    HIPCHK(hipGetLastError());
  } else if (broadcast) {
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipMemcpy(b->ctrl_row, ctrl, sizeof(double) * nu, hipMemcpyHostToDevice));
    int total = b->n * nu;
    hipLaunchKernelGGL(sg_fill_rows_kernel, dim3((total + 255) / 256), dim3(256), 0, s, b->ctrl, b->ctrl_row, b->n, nu);
    HIPCHK(hipGetLastError());
  } else {
    HIPCHK(hipMemcpyAsync(b->ctrl, ctrl, sizeof(double) * b->n * nu, hipMemcpyDeviceToDevice, s));
  }
  return SG_OK;
}

// a pair of events registered in `list` BEFORE anything is recorded (so that a failing HIP call cannot leak them), first one recorded
static int begin_event_pair(sg_batch* b, std::vector<std::pair<hipEvent_t, hipEvent_t>>& list, hipStream_t s) {
  hipEvent_t e[2];
  for (int i = 0; i < 2; i++) {
    if (!b->ev_pool.empty()) { e[i] = b->ev_pool.back(); b->ev_pool.pop_back(); }
    else if (hipEventCreate(&e[i]) != hipSuccess) {
      if (i == 1) b->ev_pool.push_back(e[0]);
      return fail(SG_ERR_HIP, "hipEventCreate failed");
    }
  }
  list.emplace_back(e[0], e[1]);
  HIPCHK(hipEventRecord(e[0], s));
  return SG_OK;
}

// envs per solver wavefront (sg_pgs_rows_kernel<.., EPW>): 8 when that already gives every SIMD of the chip a wavefront
// (SG_EPW8_MIN_WAVES wavefronts of 8 envs), else 4; sg_set_solver_envs_per_wavefront / the env var SG_PGS_EPW override;
// neighbour-row models always run 4 envs of 16 lanes
#define SG_EPW8_MIN_WAVES 1024
static int solver_epw(const sg_batch* b) {
  if (b->m->plan.h.nnb > 0) return 4;
  if (b->epw_override) return b->epw_override;
  if (const char* pe = getenv("SG_PGS_EPW")) return atoi(pe) == 4 ? 4 : 8;
  return (b->n + 7) / 8 >= SG_EPW8_MIN_WAVES ? 8 : 4;
}

// split pipeline: phase(begin) -> pgs -> [phase(finish+begin) -> pgs]* -> phase(finish)
static int launch_split(sg_batch* b, int mode, const uint8_t* mask, int nsub, double* sens, long long stride, int32_t* flags, int32_t* touch,
                        hipStream_t s) {
  const SgPlanHeader& H = b->m->plan.h;
  SgPhaseArgs pa;
  pa.H = b->dH; pa.elem = b->delem;
  pa.qpos = b->qpos; pa.qvel = b->qvel; pa.warm = b->warm; pa.act = b->act; pa.ctrl = b->ctrl;
  pa.kenv = b->kenv; pa.kmask_jnt = b->kmask_jnt; pa.kmask_ten = b->kmask_ten;
  pa.mask = mask; pa.sens = nullptr; pa.sens_stride = stride > 0 ? stride : H.nsensordata;
  pa.w = b->w; pa.nenv = b->n; pa.rowlayout = b->pipeline == 2;
  pa.nbtab = b->dnbtab; pa.cpos = b->dcpos; pa.cst_rounds = (H.nnb > 0 && SG_ROWS_NB_MODE(H.nelem, H.eq_rounds) == 2) ? H.eq_rounds + 8 : 0;
  pa.gpairs = b->dgpairs;
  pa.nelem = H.nelem; pa.nv = H.nv; pa.nu = H.nu; pa.elem_dof0 = H.elem_dof0; pa.nchain = H.nchain; pa.t0_id = H.t0_id; pa.timestep = H.timestep;
  SgPgsArgs ga;
  ga.sched = b->dsched; ga.nbtab = b->dnbtab; ga.tab = b->dtab;
  ga.H = b->dH; ga.elem = b->delem; ga.w = b->w; ga.nenv = b->n;
  const size_t lds = sizeof(double) * ((size_t)(5 * 8 + 2) * H.nelem + 16 * 4 * SG_MAXLIM + 72);   // (split pipeline, test builds)
  (void)lds;
  // rows kernel: joint-fix rows per lane (template parameter, the smallest instantiated value >= ceil(nelem / 8)); its LDS
  // arrays are padded to 8 * NSL rows
  const int nsl = sg_rows_nsl(H.nelem);
  const bool nbm = H.nnb > 0;
  const int epw = solver_epw(b);
  const int nbmode = nbm ? SG_ROWS_NB_MODE(H.nelem, H.eq_rounds) : 0;
  const size_t lds_rows = sizeof(double) * (nbm ? SG_ROWS_LDS_NB(epw, H.nelem, H.eq_rounds, nbmode == 2) : SG_ROWS_LDS_FIX(epw, 8 * (size_t)nsl));
  if (!b->lds_attr_set) {  // per device: a batch on another GPU of the same process needs its own call
    HIPCHK(sg_rows_prepare());
#ifdef SG_LEGACY_PIPELINES
    HIPCHK(sg_legacy_prepare());
#endif
    b->lds_attr_set = true;
  }
  // forward passes to run: (mode 1: one non-integrating forward first) + nsub integrating ones
  const int nfwd = nsub + (mode == 1 ? 1 : 0);
  size_t call_ev = 0;
  if (b->prof) {
    if (int rc = begin_event_pair(b, b->ev, s)) return rc;
    call_ev = b->ev.size() - 1;
  }
  // the main pass over all envs, then (rows pipeline, forward passes only) the general contact pass: a small fixed grid whose
  // its blocks stride over W.gen_list, the envs the main pass has put there -- normally none, and then every block returns at once
  const bool genpass = b->pipeline == 2;
  for (int k = 0; k <= nfwd; k++) {
    SgPhaseArgs p = pa;
    p.first = k == 0;
    p.do_reset = (mode == 1 && k == 0);
    p.do_finish = k > 0;
    p.finish_integrate = (mode == 1) ? (k > 1) : 1;  // forward k-1 was the non-integrating one iff mode 1 and k == 1
    p.do_begin = k < nfwd;
    p.sens = (k == nfwd) ? sens : nullptr;
    if (nfwd == 0) { p.do_begin = 0; p.do_finish = 0; }
    HIPCHK(sg_launch_chain(p, b->n, s));
    p.sens = nullptr;  // the chain kernel writes the sensors (they all sit on finger sites)
    HIPCHK(sg_launch_phase(p, b->m->rounds, nbm, genpass, b->n, s));
    if (k < nfwd) {
      if (b->prof)
        if (int rc = begin_event_pair(b, b->ev_pgs, s)) return rc;
      if (b->pipeline == 2) HIPCHK(sg_launch_rows(ga, nsl, nbmode, epw, b->n, lds_rows, s));
#ifdef SG_LEGACY_PIPELINES
      else HIPCHK(sg_launch_pgs_split(ga, b->n, lds, s));
#endif
      if (b->prof) HIPCHK(hipEventRecord(b->ev_pgs.back().second, s));
    }
  }
  if (b->prof) HIPCHK(hipEventRecord(b->ev[call_ev].second, s));
  // outputs of the call: one small kernel for the five per-env int arrays (five device-to-device copies cost 1 % of a step);
  // with a mask (masked reset) only the selected envs' entries may change
  hipLaunchKernelGGL(sg_masked_copy_kernel, dim3((b->n + 255) / 256), dim3(256), 0, s, mask, b->n, b->w.status, flags ? flags : b->flags,
                     b->w.touch, touch ? touch : b->touch, b->w.ncon, b->ncon, b->w.nefc, b->nefc, b->w.iters, b->iters);
  HIPCHK(hipGetLastError());
  return SG_OK;
}

static int launch(sg_batch* b, int mode, const uint8_t* mask, int nsub, double* sens, long long stride, int32_t* flags, int32_t* touch,
                  hipStream_t s) {
  if (b->pipeline == 3) return launch_tree(b, mode, mask, nsub, sens, stride, flags, touch, s);
  if (b->pipeline >= 1) return launch_split(b, mode, mask, nsub, sens, stride, flags, touch, s);
#ifdef SG_LEGACY_PIPELINES
  const SgPlanHeader& H = b->m->plan.h;
  SgKArgs a;
  a.H = b->dH; a.elem = b->delem;
  a.qpos = b->qpos; a.qvel = b->qvel; a.warm = b->warm; a.act = b->act; a.ctrl = b->ctrl;
  a.kenv = b->kenv; a.kmask_jnt = b->kmask_jnt; a.kmask_ten = b->kmask_ten;
  a.mask = mask;
  a.sens = sens; a.sens_stride = stride > 0 ? stride : H.nsensordata;
  a.flags = flags ? flags : b->flags; a.touch = touch ? touch : b->touch;
  a.ncon = b->ncon; a.nefc = b->nefc; a.iters = b->iters;
  a.nenv = b->n; a.nsub = nsub; a.mode = mode;
  if (b->prof)
    if (int rc = begin_event_pair(b, b->ev, s)) return rc;
  HIPCHK(sg_launch_fused(a, b->m->rounds, b->n, s));
  if (b->prof) HIPCHK(hipEventRecord(b->ev.back().second, s));
  return SG_OK;
#else
  return fail(SG_ERR_MODEL, "the fused pipeline is not part of this build (test builds only: build_native.py --legacy)");
#endif
}

int sg_reset(sg_batch* b, const uint8_t* mask, int sim_start, double* sens_out, int32_t* flags_out, int32_t* touch_out, void* stream) {
  if (!b || sim_start < 0) return fail(SG_ERR_INVALID, "sg_reset: bad argument");
  HIPCHK(hipSetDevice(b->device));
  return launch(b, 1, mask, sim_start, sens_out, 0, flags_out, touch_out, (hipStream_t)stream);
}

int sg_step(sg_batch* b, int n_substeps, double* sens_out, long long sens_stride, int32_t* flags_out, int32_t* touch_out, void* stream) {
  if (!b || n_substeps < 0) return fail(SG_ERR_INVALID, "sg_step: bad argument");
  HIPCHK(hipSetDevice(b->device));
  return launch(b, 0, nullptr, n_substeps, sens_out, sens_stride, flags_out, touch_out, (hipStream_t)stream);
}

int sg_get_state(sg_batch* b, double* qpos, double* qvel, double* act, double* warm, double* ctrl, void* stream) {
  if (!b) return fail(SG_ERR_INVALID, "sg_get_state: bad argument");
  HIPCHK(hipSetDevice(b->device));
  hipStream_t s = (hipStream_t)stream;
  const size_t n = b->n, nv = b->m->plan.h.nv, nu = b->m->plan.h.nu;
  if (qpos) HIPCHK(hipMemcpyAsync(qpos, b->qpos, sizeof(double) * n * b->m->plan.h.nq, hipMemcpyDeviceToDevice, s));
  if (qvel) HIPCHK(hipMemcpyAsync(qvel, b->qvel, sizeof(double) * n * nv, hipMemcpyDeviceToDevice, s));
  if (warm) HIPCHK(hipMemcpyAsync(warm, b->warm, sizeof(double) * n * nv, hipMemcpyDeviceToDevice, s));
  if (act && nu) HIPCHK(hipMemcpyAsync(act, b->act, sizeof(double) * n * nu, hipMemcpyDeviceToDevice, s));
  if (ctrl && nu) HIPCHK(hipMemcpyAsync(ctrl, b->ctrl, sizeof(double) * n * nu, hipMemcpyDeviceToDevice, s));
  return SG_OK;
}

int sg_set_state(sg_batch* b, const double* qpos, const double* qvel, const double* act, const double* warm, const double* ctrl, void* stream) {
  if (!b) return fail(SG_ERR_INVALID, "sg_set_state: bad argument");
  HIPCHK(hipSetDevice(b->device));
  hipStream_t s = (hipStream_t)stream;
  const size_t n = b->n, nv = b->m->plan.h.nv, nu = b->m->plan.h.nu;
  if (qpos) HIPCHK(hipMemcpyAsync(b->qpos, qpos, sizeof(double) * n * b->m->plan.h.nq, hipMemcpyDeviceToDevice, s));
  if (qvel) HIPCHK(hipMemcpyAsync(b->qvel, qvel, sizeof(double) * n * nv, hipMemcpyDeviceToDevice, s));
  if (warm) HIPCHK(hipMemcpyAsync(b->warm, warm, sizeof(double) * n * nv, hipMemcpyDeviceToDevice, s));
  if (act && nu) HIPCHK(hipMemcpyAsync(b->act, act, sizeof(double) * n * nu, hipMemcpyDeviceToDevice, s));
  if (ctrl && nu) HIPCHK(hipMemcpyAsync(b->ctrl, ctrl, sizeof(double) * n * nu, hipMemcpyDeviceToDevice, s));
  return SG_OK;
}

int sg_get_solver_stats(sg_batch* b, int32_t* ncon, int32_t* nefc, int32_t* iters, void* stream) {
  if (!b) return fail(SG_ERR_INVALID, "sg_get_solver_stats: bad argument");
  HIPCHK(hipSetDevice(b->device));
  hipStream_t s = (hipStream_t)stream;
  const size_t n = b->n;
  if (ncon) HIPCHK(hipMemcpyAsync(ncon, b->ncon, sizeof(int) * n, hipMemcpyDeviceToDevice, s));
  if (nefc) HIPCHK(hipMemcpyAsync(nefc, b->nefc, sizeof(int) * n, hipMemcpyDeviceToDevice, s));
  if (iters) HIPCHK(hipMemcpyAsync(iters, b->iters, sizeof(int) * n, hipMemcpyDeviceToDevice, s));
  return SG_OK;
}

int sg_set_pipeline(sg_batch* b, int pipeline) {
  if (!b || pipeline < 0 || pipeline > 3) return fail(SG_ERR_INVALID, "sg_set_pipeline: bad argument");
  if (pipeline == 3) {
    if (!b->m->has_tree) return fail(SG_ERR_MODEL, "sg_set_pipeline: the tree pipeline does not run this model (fix-rows-only models within its capacities)");
    HIPCHK(hipSetDevice(b->device));
    if (int rc = tree_alloc(b)) return rc;
    b->pipeline = 3;
    return SG_OK;
  }
  if (!b->m->has_fast) return fail(SG_ERR_MODEL, "sg_set_pipeline: the model is outside the two-finger class, only the tree pipeline runs it");
#ifndef SG_LEGACY_PIPELINES
  if (pipeline < 2) return fail(SG_ERR_MODEL, "sg_set_pipeline: the fused and split pipelines are not part of this build (test builds only: build_native.py --legacy)");
#endif
  if (b->m->plan.h.nnb > 0 && pipeline != 2)
    return fail(SG_ERR_MODEL, "sg_set_pipeline: the model has neighbour equality rows, which only the rows pipeline supports");
  b->pipeline = pipeline;
  return SG_OK;
}

int sg_get_touch_words(sg_batch* b, int32_t* out, int nwords, void* stream) {
  if (!b || !out || nwords < 1) return fail(SG_ERR_INVALID, "sg_get_touch_words: bad argument");
  HIPCHK(hipSetDevice(b->device));
  hipStream_t s = (hipStream_t)stream;
  HIPCHK(hipMemsetAsync(out, 0, sizeof(int32_t) * (size_t)b->n * nwords, s));
  if (b->pipeline == 3)
    HIPCHK(hipMemcpy2DAsync(out, sizeof(int32_t) * nwords, b->touch_words, sizeof(int32_t) * 2, sizeof(int32_t) * (nwords < 2 ? nwords : 2), b->n,
                            hipMemcpyDeviceToDevice, s));
  else
    HIPCHK(hipMemcpy2DAsync(out, sizeof(int32_t) * nwords, b->pipeline == 0 ? b->touch : b->w.touch, sizeof(int32_t), sizeof(int32_t), b->n,
                            hipMemcpyDeviceToDevice, s));
  return SG_OK;
}
int sg_model_nboxes(const sg_model* m) {
  if (!m) return 0;
  if (!m->has_fast) return m->tree.NG;
  int n = 0;
  for (int c = 0; c < m->plan.h.nchain; c++) n += m->plan.h.chain[c].ngeom;
  return n;
}

int sg_set_solver_envs_per_wavefront(sg_batch* b, int epw) {
  if (!b || (epw != 0 && epw != 4 && epw != 8)) return fail(SG_ERR_INVALID, "sg_set_solver_envs_per_wavefront: 0 (automatic), 4 or 8");
  if (b->m->plan.h.nnb > 0 && epw == 8)
    return fail(SG_ERR_MODEL, "sg_set_solver_envs_per_wavefront: a model with neighbour equality rows runs 4 envs of 16 lanes per wavefront");
  b->epw_override = epw;
  return SG_OK;
}
int sg_solver_envs_per_wavefront(const sg_batch* b) { return b ? solver_epw(b) : 0; }
int sg_tree_workgroups_per_cu(const sg_batch* b) {
  if (!b) return fail(SG_ERR_INVALID, "sg_tree_workgroups_per_cu: bad argument");
  if (!b->tree_ready) return 0;
  const sg_model* m = b->m;
  if (hipSetDevice(b->device) != hipSuccess) return fail(SG_ERR_NO_DEVICE, "hipSetDevice");
  if (sg_tree_prepare() != hipSuccess) return fail(SG_ERR_HIP, "sg_tree_prepare");
  return sg_tree_occupancy(m->tree.CS, sgt::lds_bytes(m->tree, m->tplan.h.nelem, m->tplan.h.has_free, m->tplan.h.nnb));
}

#ifdef SG_SECTION_PROF
// profiling build only (build_native.py --prof, scripts/section_profile.py): read and clear the per-section cycle sums
int sg_debug_sections(sg_batch* b, unsigned long long* out32) {
  if (!b || !out32) return fail(SG_ERR_INVALID, "sg_debug_sections: bad argument");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out32, b->w.secprof, sizeof(unsigned long long) * 48, hipMemcpyDeviceToHost));
  HIPCHK(hipMemset(b->w.secprof, 0, sizeof(unsigned long long) * 48));
  return SG_OK;
}
#endif

#ifdef SG_DEBUG_WORK
// debugging build only (build_native.py --ko NAME -DSG_DEBUG_WORK, scripts/dev/work_diff.py): one env's tree-pipeline work space
// (sg_tree.h: staged narrowphase records | contact rows | mass-matrix blocks | the arrays lds_carve backs with global memory), to be
// held word by word against the host emulation's after the same launch.  -> the number of doubles copied (<= cap), or a negative error
long long sg_debug_tree_work(sg_batch* b, int env, double* out, long long cap) {
  if (!b || !out || env < 0 || env >= b->n || !b->tcws) return fail(SG_ERR_INVALID, "sg_debug_tree_work: bad argument");
  const sg_model* m = b->m;
  long long n = sgt::cws_doubles(m->tree, m->tplan.h.nelem, m->tplan.h.has_free, m->tplan.h.nnb);
  if (n > cap) n = cap;
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out, b->tcws + (size_t)env * sgt::cws_doubles(m->tree, m->tplan.h.nelem, m->tplan.h.has_free, m->tplan.h.nnb), sizeof(double) * n, hipMemcpyDeviceToHost));
  return n;
}
#endif

int sg_profile_enable(sg_batch* b, int enable) {
  if (!b) return fail(SG_ERR_INVALID, "sg_profile_enable: bad argument");
  b->prof = enable != 0;
  return SG_OK;
}

int sg_profile_read_solver(sg_batch* b, int reset, double* avg_ms, long long* launches) {
  if (!b) return fail(SG_ERR_INVALID, "sg_profile_read_solver: bad argument");
  HIPCHK(hipSetDevice(b->device));
  for (auto& e : b->ev_pgs) {
    HIPCHK(hipEventSynchronize(e.second));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e.first, e.second));
    b->prof_pgs_ms += ms; b->prof_pgs_n++;
    b->ev_pool.push_back(e.first); b->ev_pool.push_back(e.second);
  }
  b->ev_pgs.clear();
  if (avg_ms) *avg_ms = b->prof_pgs_n ? b->prof_pgs_ms / b->prof_pgs_n : 0.0;
  if (launches) *launches = b->prof_pgs_n;
  if (reset) { b->prof_pgs_ms = 0; b->prof_pgs_n = 0; }
  return SG_OK;
}

int sg_profile_read(sg_batch* b, int reset, double* avg_ms, long long* launches) {
  if (!b) return fail(SG_ERR_INVALID, "sg_profile_read: bad argument");
  HIPCHK(hipSetDevice(b->device));
  for (auto& e : b->ev) {
    HIPCHK(hipEventSynchronize(e.second));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e.first, e.second));
    b->prof_ms += ms; b->prof_n++;
    b->ev_pool.push_back(e.first); b->ev_pool.push_back(e.second);
  }
  b->ev.clear();
  if (avg_ms) *avg_ms = b->prof_n ? b->prof_ms / b->prof_n : 0.0;
  if (launches) *launches = b->prof_n;
  if (reset) { b->prof_ms = 0; b->prof_n = 0; }
  return SG_OK;
}

}  // extern "C"
