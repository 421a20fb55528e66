// Library plumbing: thread-local error string, argument validation, device info.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>

#include "common.h"

namespace nrhip {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: kernel launch failed: %s", what, hipGetErrorString(e));
    return NRHIP_ERR_LAUNCH;
  }
  return NRHIP_OK;
}

static Tuning read_tuning() {
  Tuning t;
  auto env = [](const char* name) { return getenv(name); };
  auto is = [&](const char* name, char c) {
    const char* e = env(name);
    return e && e[0] == c;
  };
  t.bin_round_log2 = 23;
  if (const char* e = env("NRHIP_BIN_ROUND_LOG2")) {
    const int x = atoi(e);
    if (x >= 15 && x <= 24) t.bin_round_log2 = x;
  }
  t.bin_pairs = is("NRHIP_BIN_PAIRS", 'a') ? 1 : is("NRHIP_BIN_PAIRS", 'n') ? 0 : -1;
  t.bin_transpose = !is("NRHIP_BIN_TRANSPOSE", '0');
  t.bin_stats = env("NRHIP_BIN_STATS") != nullptr;
  t.multi_bwd_runs = !is("NRHIP_MULTI_BWD_RUNS", '0');
  t.pair_bwd_runs = !is("NRHIP_PAIR_BWD_RUNS", '0');
  t.mlp_generic = env("NRHIP_MLP_GENERIC") != nullptr;
  t.mlp_split_wgrad = env("NRHIP_MLP_SPLIT_WGRAD") != nullptr;
  t.mlp_split_bf16 = env("NRHIP_MLP_SPLIT_BF16") != nullptr;
  t.mlp_pairs = env("NRHIP_MLP_PAIRS") ? (is("NRHIP_MLP_PAIRS", '1') ? 1 : 0) : -1;
  t.mlp_pairs_train = is("NRHIP_MLP_PAIRS_TRAIN", '1');
  t.sampler_actor_inline = is("NRHIP_SAMPLER_ACTOR_INLINE", '1');
  t.sdf_render_pair = is("NRHIP_SDF_RENDER_PAIR", '1');
  return t;
}

static Tuning g_tuning = read_tuning();  // at library load

const Tuning& tuning() { return g_tuning; }

int validate_grid(const nrhip_grid* g) {
  NR_REQUIRE(g, NRHIP_ERR_INVALID_ARG, "grid descriptor is NULL");
  NR_REQUIRE(g->num_levels >= 1 && g->num_levels <= NRHIP_MAX_LEVELS, NRHIP_ERR_INVALID_ARG,
             "num_levels %d outside [1,%d]", g->num_levels, NRHIP_MAX_LEVELS);
  NR_REQUIRE(g->n_features == 1 || g->n_features == 2 || g->n_features == 4 || g->n_features == 8,
             NRHIP_ERR_UNSUPPORTED, "features_per_level %d not in {1,2,4,8}", g->n_features);
  NR_REQUIRE(g->log2_table_size >= 1 && g->log2_table_size <= 26 &&
                 (int64_t)g->num_levels << g->log2_table_size <= (int64_t)1 << 31,
             NRHIP_ERR_INVALID_ARG, "log2_table_size %d unsupported (L*T must fit 2^31 rows)", g->log2_table_size);
  NR_REQUIRE(((int64_t)g->num_levels << g->log2_table_size) * g->n_features * 4 <= (int64_t)1 << 32,
             NRHIP_ERR_UNSUPPORTED, "hash table larger than 4 GiB is not supported (32-bit byte offsets)");
  NR_REQUIRE(g->param_dtype == 0 || g->param_dtype == 1, NRHIP_ERR_INVALID_ARG, "param_dtype %d not in {0,1}",
             g->param_dtype);
  return NRHIP_OK;
}

int validate_rays(const nrhip_rays* r) {
  NR_REQUIRE(r, NRHIP_ERR_INVALID_ARG, "rays descriptor is NULL");
  NR_REQUIRE(r->n_rays >= 0 && r->n_samples >= 0, NRHIP_ERR_INVALID_ARG, "negative ray/sample count");
  NR_REQUIRE(r->sample_stride == 0 || r->sample_stride >= r->n_samples, NRHIP_ERR_INVALID_ARG, "sample_stride < n_samples");
  if (r->n_rays == 0 || r->n_samples == 0) return NRHIP_OK;
  NR_REQUIRE(r->origins && r->directions && r->pixel_area && r->starts && r->ends, NRHIP_ERR_INVALID_ARG,
             "rays descriptor has a NULL pointer");
  return NRHIP_OK;
}

}  // namespace nrhip

extern "C" const char* nrhip_last_error(void) { return nrhip::g_err; }
// 200: nrhip_rays.order, render_fwd_ex, ray_order.  300: nrhip_field grew eval_table / eval_layout (trailing), the training
// glue of train_fused.hip, nrhip_field_fwd_train_ovr, nrhip_eval_layout_*.
// 510: nrhip_encode_bwd_binned / nrhip_hashgrid_bwd_binned write fp32 gradients whatever g->param_dtype says (the fp16 form
// is nrhip_encode_bwd_binned_f16); nrhip_adam_step_many_dev (+ _workspace), nrhip_tuning_reload.
// 511: nrhip_nonfinite_check_many.
extern "C" int nrhip_version(void) { return 511; }

extern "C" int nrhip_tuning_reload(void) {
  nrhip::g_tuning = nrhip::read_tuning();
  return NRHIP_OK;
}

extern "C" int nrhip_device_info(int32_t* n_cus, int32_t* n_xcds, int64_t* hbm_bytes) {
  int dev = 0;
  hipDeviceProp_t p;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) {
    nrhip::set_error("device_info: no HIP device");
    return NRHIP_ERR_LAUNCH;
  }
  if (n_cus) *n_cus = p.multiProcessorCount;
  if (n_xcds) *n_xcds = 8;
  if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
  return NRHIP_OK;
}
