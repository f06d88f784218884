// extern "C" boundary of libmmvae_hip.so (see include/mmvae_hip.h).
#include "../../include/mmvae_hip.h"
#include "multimnist.h"
#include "mnist.h"
#include "celeba.h"
#include "coco.h"
#include "plan_base.h"
#include <cstring>
#include <exception>

const char* mmvae_error_string();

#define API_GUARD_BEGIN try {
#define API_GUARD_END                                                   \
    } catch (const std::exception& e) {                                 \
        mmvae_set_error("internal exception: %s", e.what());            \
        return MMVAE_ESTATE;                                            \
    } catch (...) {                                                     \
        mmvae_set_error("internal exception");                          \
        return MMVAE_ESTATE;                                            \
    }

static inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }


// ---- generic plan queries / bind / pack for the models built on PlanBase alone (mnist, celeba)
static long long pb_bn_floats(const PlanBase* b) { return b->bn_list.empty() ? 0 : b->bn_list.back().stat_off + 2 * b->bn_list.back().C; }
static int pb_param_info(const PlanBase* b, int i, char* name, int* ndim, int* shape, long long* offset) {
    const auto& v = b->params;
    MMVAE_REQUIRE(i >= 0 && i < (int)v.size(), "param index %d out of range", i);
    strncpy(name, v[i].name.c_str(), 127); name[127] = 0;
    *ndim = v[i].ndim;
    for (int k = 0; k < 4; ++k) shape[k] = k < v[i].ndim ? v[i].shape[k] : 1;
    *offset = v[i].offset;
    return MMVAE_OK;
}
static int pb_bn_info(const PlanBase* b, int i, char* prefix, int* channels, long long* offset) {
    MMVAE_REQUIRE(i >= 0 && i < (int)b->bn_list.size(), "bn index %d out of range", i);
    strncpy(prefix, b->bn_names[i].c_str(), 127); prefix[127] = 0;
    *channels = b->bn_list[i].C; *offset = b->bn_list[i].stat_off;
    return MMVAE_OK;
}
static int pb_bind(PlanBase* P, float* params, float* grads, float* bn_stats, long long* nbt, void* packed, float* packed_vec,
                   float* gpk, float* gpk_vec, void* desc_dev, void* gdesc_dev) {
    MMVAE_REQUIRE(P && params && grads && bn_stats && nbt && packed && packed_vec && gpk && gpk_vec && desc_dev && gdesc_dev,
                  "bind: null buffer");
    ModelBuffers& b = P->buf;
    b.params = params; b.grads = grads; b.bn_stats = bn_stats; b.bn_nbt = nbt; b.packed = (bf16*)packed; b.packed_vec = packed_vec;
    b.gpk = gpk; b.gpk_vec = gpk_vec; b.desc_dev = (PackDesc*)desc_dev; b.gdesc_dev = (PackDesc*)gdesc_dev;
    P->bound = true;
    return MMVAE_OK;
}
static int pb_pack(PlanBase* P, hipStream_t s) {
    MMVAE_TRY(check_bound(P));
    if (P->no_pack) return MMVAE_OK;
    return launch_pack(P->buf.desc_dev, P->pk.d.data(), (int)P->pk.d.size(), P->buf.params, P->buf.packed, P->buf.packed_vec, s);
}
static int pb_grad_map(PlanBase* P, int* map, hipStream_t s) {
    MMVAE_TRY(check_bound(P));
    MMVAE_REQUIRE(map != nullptr, "grad_map: null argument");
    return launch_unpack_map(P->buf.gdesc_dev, P->gk.d.data(), (int)P->gk.d.size(), P->nparams, P->gk.mat_elems, map, s);
}
#define MMVAE_PLAN_API(pfx, T, BASE)                                                                                      \
    long long mmvae_##pfx##_param_count(const T* p) { return BASE(p)->nparams; }                                          \
    int mmvae_##pfx##_num_params(const T* p) { return (int)BASE(p)->params.size(); }                                      \
    int mmvae_##pfx##_param_info(const T* p, int i, char* name, int* ndim, int* shape, long long* offset) {               \
        return pb_param_info(BASE(p), i, name, ndim, shape, offset);                                                      \
    }                                                                                                                     \
    long long mmvae_##pfx##_bn_floats(const T* p) { return pb_bn_floats(BASE(p)); }                                       \
    int mmvae_##pfx##_num_bn(const T* p) { return (int)BASE(p)->bn_list.size(); }                                         \
    int mmvae_##pfx##_bn_info(const T* p, int i, char* prefix, int* channels, long long* offset) {                        \
        return pb_bn_info(BASE(p), i, prefix, channels, offset);                                                          \
    }                                                                                                                     \
    long long mmvae_##pfx##_packed_elems(const T* p) { return BASE(p)->pk.mat_elems; }                                    \
    long long mmvae_##pfx##_packed_vec_elems(const T* p) { return BASE(p)->pk.vec_elems > 0 ? BASE(p)->pk.vec_elems : 64; } \
    long long mmvae_##pfx##_gpk_elems(const T* p) { return BASE(p)->gk.mat_elems; }                                       \
    long long mmvae_##pfx##_gpk_vec_elems(const T* p) { return BASE(p)->gk.vec_elems > 0 ? BASE(p)->gk.vec_elems : 64; }  \
    size_t mmvae_##pfx##_desc_bytes(const T* p, int which) {                                                              \
        return sizeof(PackDesc) * (which == 0 ? BASE(p)->pk.d.size() : BASE(p)->gk.d.size());                             \
    }                                                                                                                     \
    int mmvae_##pfx##_desc_copy(const T* p, int which, void* host_out) {                                                  \
        memcpy(host_out, which == 0 ? BASE(p)->pk.d.data() : BASE(p)->gk.d.data(), mmvae_##pfx##_desc_bytes(p, which));   \
        return MMVAE_OK;                                                                                                  \
    }                                                                                                                     \
    size_t mmvae_##pfx##_workspace_bytes(const T* p) { return BASE(p)->ws_bytes; }                                        \
    size_t mmvae_##pfx##_module_workspace_bytes(const T* p) {                                                             \
        return BASE(p)->ws_bytes_module ? BASE(p)->ws_bytes_module : BASE(p)->ws_bytes;                                   \
    }                                                                                                                     \
    int mmvae_##pfx##_bind(T* p, float* params, float* grads, float* bn_stats, long long* nbt, void* packed,              \
                           float* packed_vec, float* gpk, float* gpk_vec, void* desc_dev, void* gdesc_dev) {              \
        API_GUARD_BEGIN                                                                                                   \
        return pb_bind(BASE(p), params, grads, bn_stats, nbt, packed, packed_vec, gpk, gpk_vec, desc_dev, gdesc_dev);     \
        API_GUARD_END                                                                                                     \
    }                                                                                                                     \
    int mmvae_##pfx##_pack_weights(T* p, void* stream) {                                                                  \
        API_GUARD_BEGIN                                                                                                   \
        return pb_pack(BASE(p), S(stream));                                                                               \
        API_GUARD_END                                                                                                     \
    }                                                                                                                     \
    int mmvae_##pfx##_grad_map(T* p, int* map, void* stream) {                                                            \
        API_GUARD_BEGIN                                                                                                   \
        return pb_grad_map(BASE(p), map, S(stream));                                                                      \
        API_GUARD_END                                                                                                     \
    }
static inline PlanBase* mnist_b(const mmvae_mnist_t* p) { return mnist_base(const_cast<mmvae_mnist_t*>(p)); }
static inline PlanBase* celeba_b(const mmvae_celeba_t* p) { return celeba_base(const_cast<mmvae_celeba_t*>(p)); }
static inline PlanBase* coco_b(const mmvae_coco_t* p) { return coco_base(const_cast<mmvae_coco_t*>(p)); }

extern "C" {

const char* mmvae_last_error(void) { return mmvae_error_string(); }
const char* mmvae_version(void) { return "mmvae-hip 0.1 (gfx950)"; }

int mmvae_init(int device) {
    API_GUARD_BEGIN
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) { mmvae_set_error("no HIP device visible"); return MMVAE_EHIP; }
    MMVAE_REQUIRE(device >= 0 && device < n, "device %d out of range (%d visible)", device, n);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { mmvae_set_error("hipGetDeviceProperties failed"); return MMVAE_EHIP; }
    MMVAE_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    if (hipSetDevice(device) != hipSuccess) { mmvae_set_error("hipSetDevice failed"); return MMVAE_EHIP; }
    return MMVAE_OK;
    API_GUARD_END
}

mmvae_mm_t* mmvae_mm_create(int n_latents, int batch) {
    try { return mm_create(n_latents, batch); } catch (...) { mmvae_set_error("mm_create failed"); return nullptr; }
}
void mmvae_mm_destroy(mmvae_mm_t* p) { mm_destroy(p); }
long long mmvae_mm_param_count(const mmvae_mm_t* p) { return mm_param_count(p); }
int mmvae_mm_num_params(const mmvae_mm_t* p) { return (int)mm_params(p).size(); }
int mmvae_mm_param_info(const mmvae_mm_t* p, int i, char* name, int* ndim, int* shape, long long* offset) {
    const auto& v = mm_params(p);
    MMVAE_REQUIRE(i >= 0 && i < (int)v.size(), "param index %d out of range", i);
    strncpy(name, v[i].name.c_str(), 127); name[127] = 0;
    *ndim = v[i].ndim;
    for (int k = 0; k < 4; ++k) shape[k] = k < v[i].ndim ? v[i].shape[k] : 1;
    *offset = v[i].offset;
    return MMVAE_OK;
}
long long mmvae_mm_bn_floats(const mmvae_mm_t* p) { return mm_bn_floats(p); }
int mmvae_mm_num_bn(const mmvae_mm_t* p) { return mm_num_bn(p); }
int mmvae_mm_bn_info(const mmvae_mm_t* p, int i, char* prefix, int* channels, long long* offset) {
    std::string s; int c; long long o;
    MMVAE_TRY(mm_bn_info(p, i, s, c, o));
    strncpy(prefix, s.c_str(), 127); prefix[127] = 0; *channels = c; *offset = o;
    return MMVAE_OK;
}
long long mmvae_mm_packed_elems(const mmvae_mm_t* p) { return mm_packed_elems(p); }
long long mmvae_mm_packed_vec_elems(const mmvae_mm_t* p) { return mm_packed_vec_elems(p); }
long long mmvae_mm_gpk_elems(const mmvae_mm_t* p) { return mm_gpk_elems(p); }
long long mmvae_mm_gpk_vec_elems(const mmvae_mm_t* p) { return mm_gpk_vec_elems(p); }
size_t mmvae_mm_desc_bytes(const mmvae_mm_t* p, int which) {
    return sizeof(PackDesc) * (size_t)(which == 0 ? mm_ndesc(p) : mm_ngdesc(p));
}
int mmvae_mm_desc_copy(const mmvae_mm_t* p, int which, void* host_out) {
    memcpy(host_out, which == 0 ? mm_desc_host(p) : mm_gdesc_host(p), mmvae_mm_desc_bytes(p, which));
    return MMVAE_OK;
}
size_t mmvae_mm_workspace_bytes(const mmvae_mm_t* p) { return mm_workspace_bytes(p); }
size_t mmvae_mm_module_workspace_bytes(const mmvae_mm_t* p) { return mm_module_workspace_bytes(p); }
int mmvae_mm_bind(mmvae_mm_t* p, float* params, float* grads, float* bn_stats, long long* nbt, void* packed, float* packed_vec,
                  float* gpk, float* gpk_vec, void* desc_dev, void* gdesc_dev) {
    API_GUARD_BEGIN
    MMBuffers b;
    b.params = params; b.grads = grads; b.bn_stats = bn_stats; b.bn_nbt = nbt; b.packed = (bf16*)packed; b.packed_vec = packed_vec;
    b.gpk = gpk; b.gpk_vec = gpk_vec; b.desc_dev = (PackDesc*)desc_dev; b.gdesc_dev = (PackDesc*)gdesc_dev;
    return mm_bind(p, b);
    API_GUARD_END
}
int mmvae_mm_grad_map(mmvae_mm_t* p, int* map, void* stream) {
    API_GUARD_BEGIN
    MMVAE_REQUIRE(p && map, "mmvae_mm_grad_map: null argument");
    return mm_grad_map(p, map, S(stream));
    API_GUARD_END
}
int mmvae_mm_pack_weights(mmvae_mm_t* p, void* stream) {
    API_GUARD_BEGIN
    return mm_pack_weights(p, S(stream));
    API_GUARD_END
}
int mmvae_mm_wait_early_grads(mmvae_mm_t* p, void* stream) {
    API_GUARD_BEGIN
    return mm_wait_early_grads(p, S(stream));
    API_GUARD_END
}
int mmvae_mm_step(mmvae_mm_t* p, const mmvae_mm_step_io* io, int training, int do_backward, void* stream) {
    API_GUARD_BEGIN
    MMVAE_REQUIRE(p && io, "mmvae_mm_step: null argument");
    MMStepIO s;
    s.ws = io->ws; s.ws_bytes = io->ws_bytes; s.step_ctr = io->step_counter; s.image = io->image; s.text = io->text; s.eps = io->eps;
    s.enc_mask1 = io->enc_mask1; s.enc_mask2 = io->enc_mask2; s.gru_keep = io->gru_keep;
    s.enc_dropout = io->enc_dropout; s.gru_dropout = io->gru_dropout; s.force_tokens = io->force_tokens;
    s.kl_lambda = io->kl_lambda;
    for (int k = 0; k < 3; ++k) { s.lambda_xy[k] = io->lambda_xy[k]; s.lambda_yx[k] = io->lambda_yx[k]; }
    s.seed = io->seed; s.sums = io->sums; s.recon_image = io->recon_image; s.recon_text = io->recon_text;
    s.mu = io->mu; s.logvar = io->logvar; s.tokens = io->tokens;
    for (int k = 0; k < 3; ++k) s.pass_skip[k] = io->pass_skip[k];
    s.defer_unpack = io->defer_unpack;
    s.pack_first = io->pack_first;
    s.dp_split = io->dp_split;
    if (io->early_adam) {
        const mmvae_early_adam& e = *io->early_adam;
        MMVAE_REQUIRE(e.m && e.v && e.state && e.gmap && e.ran, "mmvae_mm_step: early_adam with a null field");
        s.early_adam = true;
        s.ea_m = e.m; s.ea_v = e.v; s.ea_state = e.state; s.ea_lr = e.lr; s.ea_b1 = e.beta1; s.ea_b2 = e.beta2; s.ea_eps = e.eps;
        s.ea_scale = e.grad_scale; s.ea_gmap = e.gmap; s.ea_ran = e.ran;
        *e.ran = 0;
    }
    return mm_step_fwd_bwd(p, s, training, do_backward, S(stream));
    API_GUARD_END
}
int mmvae_mm_image_encoder_fwd(mmvae_mm_t* p, void* ws, size_t wsb, const float* image, const uint8_t* m1, const uint8_t* m2,
                               int training, float* out, void* stream) {
    API_GUARD_BEGIN
    return mm_image_encoder_fwd(p, ws, wsb, image, m1, m2, training, out, S(stream));
    API_GUARD_END
}
int mmvae_mm_image_encoder_bwd(mmvae_mm_t* p, void* ws, size_t wsb, const float* d_out, const uint8_t* m1, const uint8_t* m2, void* stream) {
    API_GUARD_BEGIN
    return mm_image_encoder_bwd(p, ws, wsb, d_out, m1, m2, S(stream));
    API_GUARD_END
}
int mmvae_mm_image_decoder_fwd(mmvae_mm_t* p, void* ws, size_t wsb, const float* z, int training, float* recon, void* stream) {
    API_GUARD_BEGIN
    return mm_image_decoder_fwd(p, ws, wsb, z, training, recon, S(stream));
    API_GUARD_END
}
int mmvae_mm_image_decoder_bwd(mmvae_mm_t* p, void* ws, size_t wsb, const float* d_recon, const float* recon, float* dz, void* stream) {
    API_GUARD_BEGIN
    return mm_image_decoder_bwd(p, ws, wsb, d_recon, recon, dz, S(stream));
    API_GUARD_END
}
int mmvae_mm_text_encoder_fwd(mmvae_mm_t* p, void* ws, size_t wsb, const long long* text, float* out, void* stream) {
    API_GUARD_BEGIN
    return mm_text_encoder_fwd(p, ws, wsb, text, out, S(stream));
    API_GUARD_END
}
int mmvae_mm_text_encoder_bwd(mmvae_mm_t* p, void* ws, size_t wsb, const long long* text, const float* d_out, void* stream) {
    API_GUARD_BEGIN
    return mm_text_encoder_bwd(p, ws, wsb, text, d_out, S(stream));
    API_GUARD_END
}
int mmvae_mm_text_decoder_fwd(mmvae_mm_t* p, void* ws, size_t wsb, const float* z, int training, const uint8_t* keep,
                              const long long* force_tokens, float* words, long long* tokens, void* stream) {
    API_GUARD_BEGIN
    return mm_text_decoder_fwd(p, ws, wsb, z, training, keep, force_tokens, words, tokens, S(stream));
    API_GUARD_END
}
int mmvae_mm_text_decoder_bwd(mmvae_mm_t* p, void* ws, size_t wsb, const float* z, const uint8_t* keep, const long long* force_tokens,
                              const float* words, const long long* tokens, const float* d_words, float* dz, void* stream) {
    API_GUARD_BEGIN
    return mm_text_decoder_bwd(p, ws, wsb, z, keep, force_tokens, words, tokens, d_words, dz, S(stream));
    API_GUARD_END
}
int mmvae_mm_bench_layer(mmvae_mm_t* p, void* ws, size_t wsb, const char* layer, int iters, void* stream) {
    API_GUARD_BEGIN
    return mm_bench_layer(p, ws, wsb, layer, iters, S(stream));
    API_GUARD_END
}
double mmvae_mm_layer_flops(const mmvae_mm_t* p, const char* layer) { return mm_layer_flops(p, layer); }
double mmvae_mm_layer_algo_flops(const mmvae_mm_t* p, const char* layer) { return mm_layer_algo_flops(p, layer); }
double mmvae_mm_layer_algo_bytes(const mmvae_mm_t* p, const char* layer) { return mm_layer_algo_bytes(p, layer); }
long long mmvae_mm_debug_offset(mmvae_mm_t* p, const char* name) { return mm_debug_offset(p, name); }

// ---- MNIST (mnist/model.py, mnist/train.py)
mmvae_mnist_t* mmvae_mnist_create(int n_latents, int batch) {
    try { return mnist_create(n_latents, batch); } catch (...) { mmvae_set_error("mnist_create failed"); return nullptr; }
}
mmvae_mnist_t* mmvae_mnist_create_p(int n_latents, int batch, int precision) {
    try { return mnist_create(n_latents, batch, precision); } catch (...) { mmvae_set_error("mnist_create failed"); return nullptr; }
}
int mmvae_mnist_precision(const mmvae_mnist_t* p) { return mnist_is_f32(p) ? 0 : 1; }
void mmvae_mnist_destroy(mmvae_mnist_t* p) { mnist_destroy(p); }
MMVAE_PLAN_API(mnist, mmvae_mnist_t, mnist_b)
int mmvae_mnist_step(mmvae_mnist_t* p, const mmvae_mnist_step_io* io, int training, int do_backward, void* stream) {
    API_GUARD_BEGIN
    MMVAE_REQUIRE(p && io, "mmvae_mnist_step: null argument");
    MnistStepIO s;
    s.ws = io->ws; s.ws_bytes = io->ws_bytes; s.step_ctr = io->step_counter; s.image = io->image; s.label = io->label; s.eps = io->eps;
    for (int k = 0; k < 3; ++k) { s.lambda_xy[k] = io->lambda_xy[k]; s.lambda_yx[k] = io->lambda_yx[k]; }
    s.kl_coef = io->kl_coef; s.seed = io->seed; s.sums = io->sums; s.recon_image = io->recon_image; s.recon_text = io->recon_text;
    s.mu = io->mu; s.logvar = io->logvar;
    for (int k = 0; k < 3; ++k) s.pass_skip[k] = io->pass_skip[k];
    return mnist_step(p, s, training, do_backward, S(stream));
    API_GUARD_END
}

#define MNIST_FWD(name, in_t)                                                                                             \
    int mmvae_mnist_##name##_fwd(mmvae_mnist_t* p, void* ws, size_t wsb, const in_t* in, int training, float* out, void* st) { \
        API_GUARD_BEGIN                                                                                                   \
        return mnist_##name##_fwd(p, ws, wsb, in, training, out, S(st));                                                  \
        API_GUARD_END                                                                                                     \
    }
MNIST_FWD(image_encoder, float)
MNIST_FWD(image_decoder, float)
MNIST_FWD(text_encoder, long long)
MNIST_FWD(text_decoder, float)
int mmvae_mnist_image_encoder_bwd(mmvae_mnist_t* p, void* ws, size_t wsb, const float* d_out, void* st) {
    API_GUARD_BEGIN
    return mnist_image_encoder_bwd(p, ws, wsb, d_out, S(st));
    API_GUARD_END
}
int mmvae_mnist_image_decoder_bwd(mmvae_mnist_t* p, void* ws, size_t wsb, const float* d_recon, const float* recon, float* dz, void* st) {
    API_GUARD_BEGIN
    return mnist_image_decoder_bwd(p, ws, wsb, d_recon, recon, dz, S(st));
    API_GUARD_END
}
int mmvae_mnist_text_encoder_bwd(mmvae_mnist_t* p, void* ws, size_t wsb, const long long* label, const float* d_out, void* st) {
    API_GUARD_BEGIN
    return mnist_text_encoder_bwd(p, ws, wsb, label, d_out, S(st));
    API_GUARD_END
}
int mmvae_mnist_text_decoder_bwd(mmvae_mnist_t* p, void* ws, size_t wsb, const float* d_logp, const float* logp, float* dz, void* st) {
    API_GUARD_BEGIN
    return mnist_text_decoder_bwd(p, ws, wsb, d_logp, logp, dz, S(st));
    API_GUARD_END
}

// ---- CelebA (celeba/model.py, celeba/train.py)
mmvae_celeba_t* mmvae_celeba_create(int n_latents, int batch) {
    try { return celeba_create(n_latents, batch); } catch (...) { mmvae_set_error("celeba_create failed"); return nullptr; }
}
void mmvae_celeba_destroy(mmvae_celeba_t* p) { celeba_destroy(p); }
MMVAE_PLAN_API(celeba, mmvae_celeba_t, celeba_b)
int mmvae_celeba_step(mmvae_celeba_t* p, const mmvae_celeba_step_io* io, int training, int do_backward, void* stream) {
    API_GUARD_BEGIN
    MMVAE_REQUIRE(p && io, "mmvae_celeba_step: null argument");
    CelebaStepIO s;
    s.ws = io->ws; s.ws_bytes = io->ws_bytes; s.step_ctr = io->step_counter; s.image = io->image; s.attrs = io->attrs; s.eps = io->eps;
    s.enc_mask = io->enc_mask; s.enc_dropout = io->enc_dropout; s.kl_lambda = io->kl_lambda;
    for (int k = 0; k < 3; ++k) { s.lambda_x[k] = io->lambda_x[k]; s.lambda_y[k] = io->lambda_y[k]; }
    s.seed = io->seed; s.sums = io->sums; s.recon_image = io->recon_image; s.recon_attrs = io->recon_attrs;
    s.mu = io->mu; s.logvar = io->logvar;
    for (int k = 0; k < 3; ++k) s.pass_skip[k] = io->pass_skip[k];
    s.defer_unpack = io->defer_unpack;
    return celeba_step(p, s, training, do_backward, S(stream));
    API_GUARD_END
}
int mmvae_celeba_image_encoder_fwd(mmvae_celeba_t* p, void* ws, size_t wsb, const float* image, const uint8_t* mask, int training, float* out, void* st) {
    API_GUARD_BEGIN
    return celeba_image_encoder_fwd(p, ws, wsb, image, mask, training, out, S(st));
    API_GUARD_END
}
int mmvae_celeba_image_encoder_bwd(mmvae_celeba_t* p, void* ws, size_t wsb, const float* d_out, const uint8_t* mask, void* st) {
    API_GUARD_BEGIN
    return celeba_image_encoder_bwd(p, ws, wsb, d_out, mask, S(st));
    API_GUARD_END
}
int mmvae_celeba_image_decoder_fwd(mmvae_celeba_t* p, void* ws, size_t wsb, const float* z, int training, float* recon, void* st) {
    API_GUARD_BEGIN
    return celeba_image_decoder_fwd(p, ws, wsb, z, training, recon, S(st));
    API_GUARD_END
}
int mmvae_celeba_image_decoder_bwd(mmvae_celeba_t* p, void* ws, size_t wsb, const float* d_recon, const float* recon, float* dz, void* st) {
    API_GUARD_BEGIN
    return celeba_image_decoder_bwd(p, ws, wsb, d_recon, recon, dz, S(st));
    API_GUARD_END
}
int mmvae_celeba_attrs_encoder_fwd(mmvae_celeba_t* p, void* ws, size_t wsb, const float* attrs, int training, float* out, void* st) {
    API_GUARD_BEGIN
    return celeba_attrs_encoder_fwd(p, ws, wsb, attrs, training, out, S(st));
    API_GUARD_END
}
int mmvae_celeba_attrs_encoder_bwd(mmvae_celeba_t* p, void* ws, size_t wsb, const float* d_out, void* st) {
    API_GUARD_BEGIN
    return celeba_attrs_encoder_bwd(p, ws, wsb, d_out, S(st));
    API_GUARD_END
}
int mmvae_celeba_attrs_decoder_fwd(mmvae_celeba_t* p, void* ws, size_t wsb, const float* z, int training, float* recon, void* st) {
    API_GUARD_BEGIN
    return celeba_attrs_decoder_fwd(p, ws, wsb, z, training, recon, S(st));
    API_GUARD_END
}
int mmvae_celeba_attrs_decoder_bwd(mmvae_celeba_t* p, void* ws, size_t wsb, const float* d_recon, const float* recon, float* dz, void* st) {
    API_GUARD_BEGIN
    return celeba_attrs_decoder_bwd(p, ws, wsb, d_recon, recon, dz, S(st));
    API_GUARD_END
}

// ---- COCO (coco/model.py, coco/train.py)
mmvae_coco_t* mmvae_coco_create_t(int n_latents, int batch, int steps) {
    try { return coco_create(n_latents, batch, steps); } catch (...) { mmvae_set_error("coco_create failed"); return nullptr; }
}
mmvae_coco_t* mmvae_coco_create(int n_latents, int batch) { return mmvae_coco_create_t(n_latents, batch, 102); }
void mmvae_coco_destroy(mmvae_coco_t* p) { coco_destroy(p); }
int mmvae_coco_steps(const mmvae_coco_t* p) { return p ? coco_steps(p) : 0; }
MMVAE_PLAN_API(coco, mmvae_coco_t, coco_b)
int mmvae_coco_step(mmvae_coco_t* p, const mmvae_coco_step_io* io, int training, int do_backward, void* stream) {
    API_GUARD_BEGIN
    MMVAE_REQUIRE(p && io, "mmvae_coco_step: null argument");
    CocoStepIO s;
    s.ws = io->ws; s.ws_bytes = io->ws_bytes; s.step_ctr = io->step_counter; s.image = io->image; s.text = io->text; s.sos = io->sos;
    s.eps = io->eps; s.enc_mask1 = io->enc_mask1; s.enc_mask2 = io->enc_mask2; s.gru_keep = io->gru_keep;
    s.enc_dropout = io->enc_dropout; s.gru_dropout = io->gru_dropout; s.kl_lambda = io->kl_lambda;
    for (int k = 0; k < 3; ++k) { s.lambda_xy[k] = io->lambda_xy[k]; s.lambda_yx[k] = io->lambda_yx[k]; }
    s.seed = io->seed; s.sums = io->sums; s.recon_image = io->recon_image; s.recon_text = io->recon_text;
    s.mu = io->mu; s.logvar = io->logvar;
    for (int k = 0; k < 3; ++k) s.pass_skip[k] = io->pass_skip[k];
    s.defer_unpack = io->defer_unpack; s.pack_first = io->pack_first; s.optimizer_state = io->optimizer_state;
    return coco_step(p, s, training, do_backward, S(stream));
    API_GUARD_END
}
int mmvae_coco_image_encoder_fwd(mmvae_coco_t* p, void* ws, size_t wsb, const float* image, const uint8_t* m1, const uint8_t* m2, int training, float* out, void* st) {
    API_GUARD_BEGIN
    return coco_image_encoder_fwd(p, ws, wsb, image, m1, m2, training, out, S(st));
    API_GUARD_END
}
int mmvae_coco_image_encoder_bwd(mmvae_coco_t* p, void* ws, size_t wsb, const float* d_out, const uint8_t* m1, const uint8_t* m2, void* st) {
    API_GUARD_BEGIN
    return coco_image_encoder_bwd(p, ws, wsb, d_out, m1, m2, S(st));
    API_GUARD_END
}
int mmvae_coco_image_decoder_fwd(mmvae_coco_t* p, void* ws, size_t wsb, const float* z, int training, float* recon, void* st) {
    API_GUARD_BEGIN
    return coco_image_decoder_fwd(p, ws, wsb, z, training, recon, S(st));
    API_GUARD_END
}
int mmvae_coco_image_decoder_bwd(mmvae_coco_t* p, void* ws, size_t wsb, const float* d_recon, const float* recon, float* dz, void* st) {
    API_GUARD_BEGIN
    return coco_image_decoder_bwd(p, ws, wsb, d_recon, recon, dz, S(st));
    API_GUARD_END
}
int mmvae_coco_text_encoder_fwd(mmvae_coco_t* p, void* ws, size_t wsb, const float* text, float* out, void* st) {
    API_GUARD_BEGIN
    return coco_text_encoder_fwd(p, ws, wsb, text, out, S(st));
    API_GUARD_END
}
int mmvae_coco_text_encoder_bwd(mmvae_coco_t* p, void* ws, size_t wsb, const float* text, const float* d_out, void* st) {
    API_GUARD_BEGIN
    return coco_text_encoder_bwd(p, ws, wsb, text, d_out, S(st));
    API_GUARD_END
}
int mmvae_coco_text_decoder_fwd(mmvae_coco_t* p, void* ws, size_t wsb, const float* z, const float* sos, const uint8_t* keep, int training, float* sentence, void* st) {
    API_GUARD_BEGIN
    return coco_text_decoder_fwd(p, ws, wsb, z, sos, keep, training, sentence, S(st));
    API_GUARD_END
}
int mmvae_coco_text_decoder_bwd(mmvae_coco_t* p, void* ws, size_t wsb, const float* z, const float* sos, const uint8_t* keep, const float* sentence, const float* d_sentence, float* dz, void* st) {
    API_GUARD_BEGIN
    return coco_text_decoder_bwd(p, ws, wsb, z, sos, keep, sentence, d_sentence, dz, S(st));
    API_GUARD_END
}


int mmvae_poe_fwd(const float* mu, const float* lv, int M, int n, float* omu, float* olv, void* s) { return launch_poe_fwd(mu, lv, M, n, omu, olv, S(s)); }
int mmvae_poe_bwd(const float* mu, const float* lv, int M, int n, const float* gmu, const float* glv, float* dmu, float* dlv, void* s) {
    return launch_poe_bwd(mu, lv, M, n, gmu, glv, dmu, dlv, S(s));
}
int mmvae_reparam_fwd(const float* mu, const float* lv, const float* eps, int n, float* z, void* s) { return launch_reparam_fwd(mu, lv, eps, n, z, S(s)); }
int mmvae_reparam_bwd(const float* lv, const float* eps, const float* dz, int n, float* dmu, float* dlv, void* s) {
    return launch_reparam_bwd(lv, eps, dz, n, dmu, dlv, S(s));
}
int mmvae_kl_fwd(const float* mu, const float* lv, int n, float* out, void* s) { return launch_kl_fwd(mu, lv, n, out, S(s)); }
int mmvae_kl_bwd(const float* mu, const float* lv, int n, float coef, const float* gs, float* dmu, float* dlv, void* s) { return launch_kl_bwd(mu, lv, n, coef, gs, dmu, dlv, S(s)); }
int mmvae_bce_fwd(const float* p, const float* t, long long n, float* out, void* s) { return launch_bce_fwd(p, t, n, out, S(s)); }
int mmvae_bce_bwd(const float* p, const float* t, long long n, float coef, const float* gs, float* dp, void* s) { return launch_bce_bwd(p, t, n, coef, gs, dp, S(s)); }
int mmvae_nll_fwd(const float* lp, const long long* tg, int rows, int classes, float* out, void* s) { return launch_nll_fwd(lp, tg, rows, classes, out, S(s)); }
int mmvae_nll_bwd(const long long* tg, int rows, int classes, float coef, const float* gs, float* dlp, void* s) { return launch_nll_bwd(tg, rows, classes, coef, gs, dlp, S(s)); }
int mmvae_normal(float* out, long long n, unsigned long long seed, const long long* ctr, unsigned sid, void* s) { return launch_normal(out, n, seed, ctr, sid, S(s)); }
int mmvae_keep_mask(uint8_t* out, long long n, float p, unsigned long long seed, const long long* ctr, unsigned sid, void* s) {
    return launch_keep_mask(out, n, p, seed, ctr, sid, S(s));
}
int mmvae_mse_fwd(const float* a, const float* b, long long n, float* out, void* s) { return launch_mse_fwd(a, b, n, out, S(s)); }
int mmvae_mse_bwd(const float* a, const float* b, long long n, float coef, const float* gs, float* da, void* s) { return launch_mse_bwd(a, b, n, coef, gs, da, S(s)); }
int mmvae_u8_to_f32(const uint8_t* src, long long n, float denom, float* dst, void* s) { return launch_u8_to_f32(src, n, denom, dst, S(s)); }
int mmvae_u8_to_f32_after(const uint8_t* src, long long n, float denom, float* dst, void* wait_event, void* s) {
    if (wait_event && hipStreamWaitEvent(S(s), reinterpret_cast<hipEvent_t>(wait_event), 0) != hipSuccess) {
        mmvae_set_error("u8_to_f32_after: hipStreamWaitEvent: %s", hipGetErrorString(hipGetLastError()));
        return MMVAE_EHIP;
    }
    return launch_u8_to_f32(src, n, denom, dst, S(s));
}
int mmvae_gather_rows_u8_f32(const uint8_t* src, const long long* idx, long long rows, long long row_elems, float denom, float* dst, void* s) {
    return launch_gather_rows_u8_f32(src, idx, rows, row_elems, denom, dst, S(s));
}
int mmvae_stream_wait_event(void* s, void* e) {
    MMVAE_REQUIRE(e, "stream_wait_event: null event");
    if (hipStreamWaitEvent(S(s), reinterpret_cast<hipEvent_t>(e), 0) != hipSuccess) { mmvae_set_error("hipStreamWaitEvent: %s", hipGetErrorString(hipGetLastError())); return MMVAE_EHIP; }
    return MMVAE_OK;
}
int mmvae_event_create(void** out) {
    MMVAE_REQUIRE(out, "event_create: null argument");
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) {
        mmvae_set_error("hipEventCreate failed: %s", hipGetErrorString(hipGetLastError()));
        return MMVAE_EHIP;
    }
    *out = e;
    return MMVAE_OK;
}
int mmvae_event_destroy(void* e) { return e && hipEventDestroy(reinterpret_cast<hipEvent_t>(e)) != hipSuccess ? MMVAE_EHIP : MMVAE_OK; }
int mmvae_event_record(void* e, void* s) {
    MMVAE_REQUIRE(e, "event_record: null event");
    if (hipEventRecord(reinterpret_cast<hipEvent_t>(e), S(s)) != hipSuccess) { mmvae_set_error("hipEventRecord: %s", hipGetErrorString(hipGetLastError())); return MMVAE_EHIP; }
    return MMVAE_OK;
}
int mmvae_event_synchronize(void* e) {
    MMVAE_REQUIRE(e, "event_synchronize: null event");
    if (hipEventSynchronize(reinterpret_cast<hipEvent_t>(e)) != hipSuccess) { mmvae_set_error("hipEventSynchronize: %s", hipGetErrorString(hipGetLastError())); return MMVAE_EHIP; }
    return MMVAE_OK;
}
int mmvae_h2d_stage(void* dst_a, const void* src_a, size_t bytes_a, void* dst_b, const void* src_b, size_t bytes_b, void* event, void* s) {
    MMVAE_REQUIRE(dst_a && src_a && bytes_a > 0 && event && (bytes_b == 0 || (dst_b && src_b)), "h2d_stage: null argument");
    hipError_t e = hipMemcpyAsync(dst_a, src_a, bytes_a, hipMemcpyHostToDevice, S(s));
    if (e == hipSuccess && bytes_b > 0) e = hipMemcpyAsync(dst_b, src_b, bytes_b, hipMemcpyHostToDevice, S(s));
    if (e == hipSuccess) e = hipEventRecord(reinterpret_cast<hipEvent_t>(event), S(s));
    if (e != hipSuccess) { mmvae_set_error("h2d_stage: %s", hipGetErrorString(e)); (void)hipGetLastError(); return MMVAE_EHIP; }
    return MMVAE_OK;
}
int mmvae_adam_step(float* p, const float* g, float* m, float* v, long long n, long long* state, float lr, float b1, float b2,
                    float eps, float grad_scale, void* s) {
    AdamArgs a{};
    a.p = p; a.g = g; a.m = m; a.v = v; a.n = n; a.step = state; a.lr = lr; a.b1 = b1; a.b2 = b2; a.eps = eps; a.grad_scale = grad_scale;
    return launch_adam(a, S(s));
}
int mmvae_stream_create(void** out) {
    MMVAE_REQUIRE(out, "stream_create: null argument");
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
        mmvae_set_error("hipStreamCreate failed: %s", hipGetErrorString(hipGetLastError()));
        return MMVAE_EHIP;
    }
    *out = s;
    return MMVAE_OK;
}
int mmvae_stream_destroy(void* s) {
    if (s && hipStreamDestroy(S(s)) != hipSuccess) { mmvae_set_error("hipStreamDestroy failed"); return MMVAE_EHIP; }
    return MMVAE_OK;
}
int mmvae_gather_rows(const void* src, const long long* idx, long long rows, long long row_bytes, void* dst, void* s) {
    return launch_gather_rows(src, idx, rows, row_bytes, dst, S(s));
}
int mmvae_step_status(const float* sums, void* s) {
    MMVAE_REQUIRE(sums, "step_status: null sums");
    float h[16];
    if (hipMemcpyAsync(h, sums, sizeof(h), hipMemcpyDeviceToHost, S(s)) != hipSuccess || hipStreamSynchronize(S(s)) != hipSuccess) {
        mmvae_set_error("step_status: %s", hipGetErrorString(hipGetLastError()));
        return MMVAE_EHIP;
    }
    if (h[15] != h[15]) {        // only a void step writes NaN there (no loss term accumulates into word 15)
        mmvae_set_error("the step gave up on a device-side exchange (caption decoder cluster): losses are NaN, the optimizer update was skipped");
        return MMVAE_ETIMEOUT;
    }
    return MMVAE_OK;
}
int mmvae_step_losses(const float* sums, const float* w_bce, const float* w_nll, const float* w_kl, float* losses, void* s) {
    MMVAE_REQUIRE(w_bce && w_nll && w_kl, "step_losses: null weights");
    StepLossArgs a{};
    a.sums = sums; a.out = losses;
    for (int k = 0; k < 3; ++k) { a.w_bce[k] = w_bce[k]; a.w_nll[k] = w_nll[k]; a.w_kl[k] = w_kl[k]; }
    return launch_step_losses(a, S(s));
}
int mmvae_adam_step_packed_ranges(float* p, float* g, float* m, float* v, long long n, const long long* ranges, int nr, int advance,
                                  long long* state, float lr, float b1, float b2, float eps, float grad_scale, const int* gmap,
                                  const float* gpk, const float* gpk_vec, void* s) {
    MMVAE_REQUIRE(gmap && gpk && gpk_vec && ranges && nr >= 1 && nr <= 4, "adam_step_packed_ranges: null argument or more than 4 ranges");
    AdamArgs a{};
    a.p = p; a.g = g; a.m = m; a.v = v; a.n = n; a.step = state; a.lr = lr; a.b1 = b1; a.b2 = b2; a.eps = eps; a.grad_scale = grad_scale;
    a.gmap = gmap; a.gpk = gpk; a.gpk_vec = gpk_vec; a.g_out = g;
    a.nr = nr; a.no_advance = advance ? 0 : 1;
    for (int r = 0; r < nr; ++r) { a.roff[r] = ranges[2 * r]; a.rlen[r] = ranges[2 * r + 1]; }
    return launch_adam(a, S(s));
}
int mmvae_mm_early_ranges(const mmvae_mm_t* p, long long* ranges, int cap) {
    if (!p || !ranges) return 0;
    return mm_early_ranges(p, ranges, cap);
}
int mmvae_adam_step_packed(float* p, float* g, float* m, float* v, long long n, long long* state, float lr, float b1, float b2,
                           float eps, float grad_scale, const int* gmap, const float* gpk, const float* gpk_vec, void* s) {
    MMVAE_REQUIRE(gmap && gpk && gpk_vec, "adam_step_packed: null gradient map / packed buffers");
    AdamArgs a{};
    a.p = p; a.g = g; a.m = m; a.v = v; a.n = n; a.step = state; a.lr = lr; a.b1 = b1; a.b2 = b2; a.eps = eps; a.grad_scale = grad_scale;
    a.gmap = gmap; a.gpk = gpk; a.gpk_vec = gpk_vec; a.g_out = g;
    return launch_adam(a, S(s));
}

}  // extern "C"
