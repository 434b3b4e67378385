// gpvae_ffi.cc — XLA FFI handlers over libgpvae_hip.so: the adapter a maintainer of the reference
// (JAX + NumPyro) builds to call the library from jitted code.  One handler per compute entry point of
// include/gpvae.h; every handler only unpacks shapes and forwards device pointers and XLA's stream.
//
// NOT COMPILED IN THIS REPOSITORY'S IMAGE: xla/ffi/api/ffi.h ships with jaxlib, which is absent here (no
// network).  Build where jaxlib is installed:
//   hipcc -std=c++17 -shared -fPIC -I$(python -c 'import jaxlib, os; print(os.path.join(os.path.dirname(jaxlib.__file__), "include"))') \
//         -I../../include gpvae_ffi.cc -L../../gp_vae_amd -lgpvae_hip -o libgpvae_ffi.so
// The Python side (registration, custom_vjp wrappers with the reference's signatures) is gpvae_jax.py.
//
// Contract the C ABI already meets: buffers are device memory owned by XLA for the
// duration of the call, the stream is XLA's compute stream, a handler must not synchronise or allocate.
// Workspaces come from XLA's scratch allocator (ffi::ScratchAllocator), sized by the *_workspace_bytes
// queries.  Errors of the library (negative return codes) become ffi::Error::InvalidArgument with
// gpvae_last_error() as the message.
#include <cstdint>

#include "xla/ffi/api/ffi.h"

#include "gpvae.h"
#include <hip/hip_runtime.h>

namespace ffi = xla::ffi;

namespace {

inline ffi::Error Status(int rc) {
  return rc == 0 ? ffi::Error::Success() : ffi::Error::InvalidArgument(gpvae_last_error());
}
// GPVAE_F32 / GPVAE_F64 from a buffer's element type
inline int DType(ffi::DataType t, ffi::Error *err) {
  if (t == ffi::F32) return GPVAE_F32;
  if (t == ffi::F64) return GPVAE_F64;
  *err = ffi::Error::InvalidArgument("gpvae: float32 or float64 buffers only");
  return -1;
}
inline int64_t Dim(const ffi::AnyBuffer &b, int back) {  // dimension counted from the last
  const auto d = b.dimensions();
  return (int64_t)d.size() > back ? d[d.size() - 1 - back] : 1;
}
inline int64_t Count(const ffi::AnyBuffer &b) { return (int64_t)b.element_count(); }
// scratch of `bytes` bytes (16-byte aligned by XLA) or an error
inline void *Scratch(ffi::ScratchAllocator &sa, int64_t bytes, ffi::Error *err) {
  if (bytes <= 0) return nullptr;
  auto p = sa.Allocate((size_t)bytes, 256);
  if (!p.has_value()) {
    *err = ffi::Error(ffi::ErrorCode::kResourceExhausted, "gpvae: scratch allocation failed");
    return nullptr;
  }
  return *p;
}

// ---- a1: exp_kernel2 (src/model/gp.py:44) ---------------------------------------------------------------
// x [n_x, d], z [n_z, d], var [batch], ls [batch]  ->  K [batch, n_x, n_z]
ffi::Error GramSe(hipStream_t s, ffi::AnyBuffer x, ffi::AnyBuffer z, ffi::AnyBuffer var, ffi::AnyBuffer ls,
                  double jitter, ffi::Result<ffi::AnyBuffer> K) {
  ffi::Error err = ffi::Error::Success();
  const int dt = DType(x.element_type(), &err);
  if (err.failure()) return err;
  const int64_t nx = x.dimensions()[0], nz = z.dimensions()[0];
  const int d = x.dimensions().size() > 1 ? (int)x.dimensions()[1] : 1;
  return Status(gpvae_gram_se(s, dt, d, nx, nz, x.untyped_data(), z.untyped_data(), (int)Count(var),
                              var.untyped_data(), ls.untyped_data(), jitter, K->untyped_data(), nz, nx * nz));
}

// ---- a2: agg_kernel_grid (gp.py:85) with the Uniform draws as an input --------------------------------
// pts [n_cells, m, d], var [1], ls [1] -> K [n_cells, n_cells]
ffi::Error GramAgg(hipStream_t s, ffi::AnyBuffer pts, ffi::AnyBuffer var, ffi::AnyBuffer ls, double jitter,
                   ffi::Result<ffi::AnyBuffer> K) {
  ffi::Error err = ffi::Error::Success();
  const int dt = DType(pts.element_type(), &err);
  if (err.failure()) return err;
  const auto d = pts.dimensions();
  return Status(gpvae_gram_agg(s, dt, (int)d[2], d[0], (int)d[1], pts.untyped_data(), var.untyped_data(),
                               ls.untyped_data(), jitter, K->untyped_data(), d[0]));
}

// ---- a3: batched Cholesky (+ rows X L^-T) ---------------------------------------------------------------
// A [batch, n, n] (aliased to the result L via input_output_aliases), X [batch, m, n] likewise (m may be 0)
ffi::Error Potrf(hipStream_t s, ffi::ScratchAllocator sa, ffi::AnyBuffer A, ffi::AnyBuffer X, int32_t flags,
                 ffi::Result<ffi::AnyBuffer> L, ffi::Result<ffi::AnyBuffer> XL,
                 ffi::Result<ffi::Buffer<ffi::S32>> info) {
  ffi::Error err = ffi::Error::Success();
  const int dt = DType(A.element_type(), &err);
  if (err.failure()) return err;
  const int64_t n = Dim(A, 0), batch = Count(A) / (n * n > 0 ? n * n : 1);
  const int64_t m = Count(X) > 0 ? Dim(X, 1) : 0;
  // with input_output_aliases {0: 0, 1: 1} XLA hands the same memory as A / L and X / XL; without the
  // alias the operands are copied first
  if (L->untyped_data() != A.untyped_data())
    (void)hipMemcpyAsync(L->untyped_data(), A.untyped_data(), A.size_bytes(), hipMemcpyDeviceToDevice, s);
  if (m > 0 && XL->untyped_data() != X.untyped_data())
    (void)hipMemcpyAsync(XL->untyped_data(), X.untyped_data(), X.size_bytes(), hipMemcpyDeviceToDevice, s);
  void *work = Scratch(sa, gpvae_potrf_workspace_bytes(dt, n, m, (int)batch), &err);
  if (err.failure()) return err;
  return Status(gpvae_potrf(s, dt, n, (int)batch, L->untyped_data(), n, n * n, m, m > 0 ? XL->untyped_data() : nullptr,
                            n, m * n, info->typed_data(), flags, work));
}

// ---- a3/a4: draws f = L eps, y = f + sigma eps' (gp.py:190-194) ----------------------------------------
// L [n, n], E [nb, n], E2 [nb, n], sigma [1] -> F [nb, n], Y [nb, n]
ffi::Error Draw(hipStream_t s, ffi::AnyBuffer L, ffi::AnyBuffer E, ffi::AnyBuffer E2, ffi::AnyBuffer sigma,
                ffi::Result<ffi::AnyBuffer> F, ffi::Result<ffi::AnyBuffer> Y) {
  ffi::Error err = ffi::Error::Success();
  const int dt = DType(L.element_type(), &err);
  if (err.failure()) return err;
  const int64_t n = Dim(L, 0), nb = Dim(E, 1);
  return Status(gpvae_draw(s, dt, n, nb, L.untyped_data(), n, E.untyped_data(), n, E2.untyped_data(), n,
                           sigma.untyped_data(), F->untyped_data(), n, Y->untyped_data(), n));
}
// L [nb, n, n], E, E2 [nb, n], sigma [nb] -> F, Y [nb, n]     (Predictive with hyper-parameters left None)
ffi::Error DrawBatched(hipStream_t s, ffi::AnyBuffer L, ffi::AnyBuffer E, ffi::AnyBuffer E2, ffi::AnyBuffer sigma,
                       ffi::Result<ffi::AnyBuffer> F, ffi::Result<ffi::AnyBuffer> Y) {
  ffi::Error err = ffi::Error::Success();
  const int dt = DType(L.element_type(), &err);
  if (err.failure()) return err;
  const int64_t n = Dim(L, 0), nb = Dim(E, 1);
  return Status(gpvae_draw_batched(s, dt, n, nb, L.untyped_data(), n, n * n, E.untyped_data(), n, E2.untyped_data(), n,
                                   sigma.untyped_data(), F->untyped_data(), n, Y->untyped_data(), n));
}

// ---- f3: Philox normals and hyper-prior samples -------------------------------------------------------
ffi::Error Randn(hipStream_t s, int64_t seed, int64_t stream_id, ffi::Result<ffi::AnyBuffer> out) {
  ffi::Error err = ffi::Error::Success();
  const int dt = DType(out->element_type(), &err);
  if (err.failure()) return err;
  return Status(gpvae_randn(s, dt, (int64_t)out->element_count(), (uint64_t)seed, (uint64_t)stream_id,
                            out->untyped_data()));
}
ffi::Error HyperSample(hipStream_t s, int64_t seed, int64_t stream_id, int64_t first_index, int32_t ls_prior,
                       double ls_a, double ls_b, double var_scale, double sigma_scale,
                       ffi::Result<ffi::AnyBuffer> ls, ffi::Result<ffi::AnyBuffer> var,
                       ffi::Result<ffi::AnyBuffer> sigma) {
  ffi::Error err = ffi::Error::Success();
  const int dt = DType(ls->element_type(), &err);
  if (err.failure()) return err;
  return Status(gpvae_hyper_sample(s, dt, (int64_t)ls->element_count(), (uint64_t)seed, (uint64_t)stream_id,
                                   first_index, ls_prior, ls_a, ls_b, var_scale, sigma_scale, ls->untyped_data(),
                                   var->untyped_data(), sigma->untyped_data()));
}

// ---- f4: the LGCP observation of PoiGP.sample: rate = exp(f), y ~ Poisson(rate) (gp.py:249-250) ----------
ffi::Error PoissonExp(hipStream_t s, int64_t seed, int64_t stream_id, int64_t first_index, ffi::AnyBuffer f,
                      ffi::Result<ffi::AnyBuffer> rate, ffi::Result<ffi::AnyBuffer> y) {
  ffi::Error err = ffi::Error::Success();
  const int dt = DType(f.element_type(), &err);
  if (err.failure()) return err;
  if (y->element_type() != ffi::DataType::S32) return ffi::Error::InvalidArgument("gpvae_poisson_exp: y must be int32");
  return Status(gpvae_poisson_exp(s, dt, (int64_t)f.element_count(), (uint64_t)seed, (uint64_t)stream_id, first_index,
                                  f.untyped_data(), rate->untyped_data(), (int32_t *)y->untyped_data()));
}

// ---- a9 / f2: kriging (inference.py:427-460; the vmap of 511-521 when var / ls / sigma are vectors) -----
// x_obs [n_obs, d], x_test [n_test, d], y_obs [n_obs], var, ls, sigma [batch], xi [batch, n_test]
//   -> mean, sample, var_out [batch, n_test], info [batch]
ffi::Error KrigPredict(hipStream_t s, ffi::ScratchAllocator sa, ffi::AnyBuffer x_obs, ffi::AnyBuffer x_test,
                       ffi::AnyBuffer y_obs, ffi::AnyBuffer var, ffi::AnyBuffer ls, ffi::AnyBuffer sigma,
                       ffi::AnyBuffer xi, double jitter, ffi::Result<ffi::AnyBuffer> mean,
                       ffi::Result<ffi::AnyBuffer> sample, ffi::Result<ffi::AnyBuffer> var_out,
                       ffi::Result<ffi::Buffer<ffi::S32>> info) {
  ffi::Error err = ffi::Error::Success();
  const int dt = DType(x_obs.element_type(), &err);
  if (err.failure()) return err;
  const int64_t n_obs = x_obs.dimensions()[0], n_test = x_test.dimensions()[0];
  const int d = x_obs.dimensions().size() > 1 ? (int)x_obs.dimensions()[1] : 1;
  const int batch = (int)Count(var);
  void *work = Scratch(sa, gpvae_krig_batched_workspace_bytes(dt, n_obs, n_test, batch), &err);
  if (err.failure()) return err;
  return Status(gpvae_krig_predict_batched(s, dt, d, n_obs, n_test, batch, x_obs.untyped_data(), x_test.untyped_data(),
                                           y_obs.untyped_data(), var.untyped_data(), ls.untyped_data(),
                                           sigma.untyped_data(), jitter, xi.untyped_data(), mean->untyped_data(),
                                           sample->untyped_data(), var_out->untyped_data(), info->typed_data(), work));
}

// ---- a10 / f1: MVN log-density and gradient; the whole gp_regression potential -------------------------
// x [n, d], f [batch, n], var, ls [batch] -> logp [batch], grad_f [batch, n], grad_var, grad_ls [batch], info
ffi::Error MvnLogp(hipStream_t s, ffi::ScratchAllocator sa, ffi::AnyBuffer x, ffi::AnyBuffer f, ffi::AnyBuffer var,
                   ffi::AnyBuffer ls, double jitter, ffi::Result<ffi::AnyBuffer> logp,
                   ffi::Result<ffi::AnyBuffer> grad_f, ffi::Result<ffi::AnyBuffer> grad_var,
                   ffi::Result<ffi::AnyBuffer> grad_ls, ffi::Result<ffi::Buffer<ffi::S32>> info) {
  ffi::Error err = ffi::Error::Success();
  const int dt = DType(x.element_type(), &err);
  if (err.failure()) return err;
  const int64_t n = x.dimensions()[0];
  const int d = x.dimensions().size() > 1 ? (int)x.dimensions()[1] : 1;
  const int batch = (int)Count(var);
  void *work = Scratch(sa, gpvae_mvn_workspace_bytes(dt, n, batch), &err);
  if (err.failure()) return err;
  return Status(gpvae_mvn_logp(s, dt, d, n, batch, x.untyped_data(), f.untyped_data(), var.untyped_data(),
                               ls.untyped_data(), jitter, logp->untyped_data(), grad_f->untyped_data(),
                               grad_var->untyped_data(), grad_ls->untyped_data(), info->typed_data(), work));
}
// x [n, d], q [chains, n + 3], y_lik [n_lik], lik_idx [n_lik] (int64; empty: every point) -> U [chains], dU [chains, n + 3]
ffi::Error GpPotential(hipStream_t s, ffi::ScratchAllocator sa, ffi::AnyBuffer x, ffi::AnyBuffer q,
                       ffi::AnyBuffer y_lik, ffi::Buffer<ffi::S64> lik_idx, double jitter,
                       ffi::Result<ffi::AnyBuffer> U, ffi::Result<ffi::AnyBuffer> dU,
                       ffi::Result<ffi::Buffer<ffi::S32>> info) {
  ffi::Error err = ffi::Error::Success();
  const int dt = DType(x.element_type(), &err);
  if (err.failure()) return err;
  const int64_t n = x.dimensions()[0], ldq = Dim(q, 0);
  const int d = x.dimensions().size() > 1 ? (int)x.dimensions()[1] : 1;
  const int chains = (int)(Count(q) / ldq);
  const int64_t n_lik = Count(y_lik);
  void *work = Scratch(sa, gpvae_gp_potential_workspace_bytes(dt, n, chains), &err);
  if (err.failure()) return err;
  return Status(gpvae_gp_potential(s, dt, d, n, chains, x.untyped_data(), q.untyped_data(), ldq, y_lik.untyped_data(),
                                   lik_idx.element_count() ? lik_idx.typed_data() : nullptr, n_lik, jitter,
                                   U->untyped_data(), dU->untyped_data(), ldq, info->typed_data(), work));
}

// ---- a5 / a6 / a7: ELBO (+ gradients), encoder, Adam ---------------------------------------------------
// params [P] (flat, include/gpvae.h order), Y [B, n], eps_z [B, z] -> loss [1], grads [P]
ffi::Error VaeElbo(hipStream_t s, ffi::ScratchAllocator sa, ffi::AnyBuffer params, ffi::AnyBuffer Y,
                   ffi::AnyBuffer eps_z, int32_t h0, int32_t h1, double obs_sigma, bool with_grads,
                   ffi::Result<ffi::AnyBuffer> loss, ffi::Result<ffi::AnyBuffer> grads) {
  ffi::Error err = ffi::Error::Success();
  const int dt = DType(Y.element_type(), &err);
  if (err.failure()) return err;
  const int64_t B = Dim(Y, 1), n = Dim(Y, 0);
  const int z = (int)Dim(eps_z, 0);
  void *work = Scratch(sa, gpvae_vae_workspace_bytes(dt, B, n, h0, h1, z), &err);
  if (err.failure()) return err;
  return Status(gpvae_vae_elbo(s, dt, B, n, h0, h1, z, params.untyped_data(), Y.untyped_data(), n, eps_z.untyped_data(),
                               obs_sigma, loss->untyped_data(), with_grads ? grads->untyped_data() : nullptr, work));
}
// enc_params (flat head), Y [B, n], eps_z [B, z] -> mu, sd, Z [B, z]
ffi::Error EncoderFwd(hipStream_t s, ffi::ScratchAllocator sa, ffi::AnyBuffer params, ffi::AnyBuffer Y,
                      ffi::AnyBuffer eps_z, int32_t h0, int32_t h1, ffi::Result<ffi::AnyBuffer> mu,
                      ffi::Result<ffi::AnyBuffer> sd, ffi::Result<ffi::AnyBuffer> Z) {
  ffi::Error err = ffi::Error::Success();
  const int dt = DType(Y.element_type(), &err);
  if (err.failure()) return err;
  const int64_t B = Dim(Y, 1), n = Dim(Y, 0);
  const int z = (int)Dim(eps_z, 0);
  void *work = Scratch(sa, gpvae_encoder_workspace_bytes(dt, B, n, h0, h1, z), &err);
  if (err.failure()) return err;
  return Status(gpvae_encoder_fwd(s, dt, B, n, h0, h1, z, params.untyped_data(), Y.untyped_data(), n,
                                  eps_z.untyped_data(), mu->untyped_data(), sd->untyped_data(), Z->untyped_data(), work));
}
// params, m, v aliased to the results (input_output_aliases {0: 0, 2: 1, 3: 2}); t counts from 1
ffi::Error AdamStep(hipStream_t s, ffi::AnyBuffer params, ffi::AnyBuffer grads, ffi::AnyBuffer m, ffi::AnyBuffer v,
                    int64_t t, double lr, double b1, double b2, double eps, ffi::Result<ffi::AnyBuffer> p_out,
                    ffi::Result<ffi::AnyBuffer> m_out, ffi::Result<ffi::AnyBuffer> v_out) {
  ffi::Error err = ffi::Error::Success();
  const int dt = DType(params.element_type(), &err);
  if (err.failure()) return err;
  const ffi::AnyBuffer *in[3] = {&params, &m, &v};
  void *out[3] = {p_out->untyped_data(), m_out->untyped_data(), v_out->untyped_data()};
  for (int i = 0; i < 3; ++i)
    if (out[i] != in[i]->untyped_data())
      (void)hipMemcpyAsync(out[i], in[i]->untyped_data(), in[i]->size_bytes(), hipMemcpyDeviceToDevice, s);
  return Status(gpvae_adam(s, dt, Count(params), out[0], grads.untyped_data(), out[1], out[2], t, lr, b1, b2, eps));
}

// ---- a11: decoder forward / VJP, and the whole VAEInference.regression potential ------------------------
// dec_params (flat tail V1 c1 V2 c2 V3 c3), Z [chains, z] -> F [chains, n]
ffi::Error DecoderFwd(hipStream_t s, ffi::ScratchAllocator sa, ffi::AnyBuffer dec, ffi::AnyBuffer Z, int32_t h0,
                      int32_t h1, ffi::Result<ffi::AnyBuffer> F) {
  ffi::Error err = ffi::Error::Success();
  const int dt = DType(Z.element_type(), &err);
  if (err.failure()) return err;
  const int64_t chains = Dim(Z, 1), n = Dim(*F, 0);
  const int z = (int)Dim(Z, 0);
  void *work = Scratch(sa, gpvae_decoder_workspace_bytes(dt, chains, n, h0, h1, z), &err);
  if (err.failure()) return err;
  return Status(gpvae_decoder_fwd(s, dt, chains, n, h0, h1, z, dec.untyped_data(), Z.untyped_data(), F->untyped_data(), work));
}
ffi::Error DecoderVjp(hipStream_t s, ffi::ScratchAllocator sa, ffi::AnyBuffer dec, ffi::AnyBuffer Z, ffi::AnyBuffer dF,
                      int32_t h0, int32_t h1, ffi::Result<ffi::AnyBuffer> dZ) {
  ffi::Error err = ffi::Error::Success();
  const int dt = DType(Z.element_type(), &err);
  if (err.failure()) return err;
  const int64_t chains = Dim(Z, 1), n = Dim(dF, 0);
  const int z = (int)Dim(Z, 0);
  void *work = Scratch(sa, gpvae_decoder_workspace_bytes(dt, chains, n, h0, h1, z), &err);
  if (err.failure()) return err;
  return Status(gpvae_decoder_vjp(s, dt, chains, n, h0, h1, z, dec.untyped_data(), Z.untyped_data(), dF.untyped_data(),
                                  dZ->untyped_data(), work));
}
// dec, q [chains, z + 1], y_obs [n_obs], obs_idx [n_obs] (int64) -> U [chains], dU [chains, z + 1]
ffi::Error VaePotential(hipStream_t s, ffi::AnyBuffer dec, ffi::AnyBuffer q, ffi::AnyBuffer y_obs,
                        ffi::Buffer<ffi::S64> obs_idx, int64_t n, int32_t h0, int32_t h1, double noise_scale,
                        ffi::Result<ffi::AnyBuffer> U, ffi::Result<ffi::AnyBuffer> dU) {
  ffi::Error err = ffi::Error::Success();
  const int dt = DType(q.element_type(), &err);
  if (err.failure()) return err;
  const int64_t ldq = Dim(q, 0), chains = Count(q) / ldq;
  const int z = (int)ldq - 1;
  if (!gpvae_vae_potential_supported(dt, h0, h1, z, Count(y_obs)))
    return ffi::Error::InvalidArgument("gpvae_vae_potential: decoder head too large for one launch; compose decoder_fwd / decoder_vjp");
  return Status(gpvae_vae_potential(s, dt, chains, n, h0, h1, z, dec.untyped_data(), q.untyped_data(), ldq,
                                    y_obs.untyped_data(), obs_idx.typed_data(), Count(y_obs), noise_scale,
                                    U->untyped_data(), dU->untyped_data(), ldq));
}

}  // namespace

#define ARG ffi::AnyBuffer
#define RET ffi::AnyBuffer
using Stream = ffi::PlatformStream<hipStream_t>;

XLA_FFI_DEFINE_HANDLER_SYMBOL(gpvae_gram_se_ffi, GramSe,
    ffi::Ffi::Bind().Ctx<Stream>().Arg<ARG>().Arg<ARG>().Arg<ARG>().Arg<ARG>().Attr<double>("jitter").Ret<RET>());
XLA_FFI_DEFINE_HANDLER_SYMBOL(gpvae_gram_agg_ffi, GramAgg,
    ffi::Ffi::Bind().Ctx<Stream>().Arg<ARG>().Arg<ARG>().Arg<ARG>().Attr<double>("jitter").Ret<RET>());
XLA_FFI_DEFINE_HANDLER_SYMBOL(gpvae_potrf_ffi, Potrf,
    ffi::Ffi::Bind().Ctx<Stream>().Ctx<ffi::ScratchAllocator>().Arg<ARG>().Arg<ARG>().Attr<int32_t>("flags")
        .Ret<RET>().Ret<RET>().Ret<ffi::Buffer<ffi::S32>>());
XLA_FFI_DEFINE_HANDLER_SYMBOL(gpvae_draw_ffi, Draw,
    ffi::Ffi::Bind().Ctx<Stream>().Arg<ARG>().Arg<ARG>().Arg<ARG>().Arg<ARG>().Ret<RET>().Ret<RET>());
XLA_FFI_DEFINE_HANDLER_SYMBOL(gpvae_draw_batched_ffi, DrawBatched,
    ffi::Ffi::Bind().Ctx<Stream>().Arg<ARG>().Arg<ARG>().Arg<ARG>().Arg<ARG>().Ret<RET>().Ret<RET>());
XLA_FFI_DEFINE_HANDLER_SYMBOL(gpvae_randn_ffi, Randn,
    ffi::Ffi::Bind().Ctx<Stream>().Attr<int64_t>("seed").Attr<int64_t>("stream_id").Ret<RET>());
XLA_FFI_DEFINE_HANDLER_SYMBOL(gpvae_hyper_sample_ffi, HyperSample,
    ffi::Ffi::Bind().Ctx<Stream>().Attr<int64_t>("seed").Attr<int64_t>("stream_id").Attr<int64_t>("first_index")
        .Attr<int32_t>("ls_prior").Attr<double>("ls_a").Attr<double>("ls_b").Attr<double>("var_scale")
        .Attr<double>("sigma_scale").Ret<RET>().Ret<RET>().Ret<RET>());
XLA_FFI_DEFINE_HANDLER_SYMBOL(gpvae_poisson_exp_ffi, PoissonExp,
    ffi::Ffi::Bind().Ctx<Stream>().Attr<int64_t>("seed").Attr<int64_t>("stream_id").Attr<int64_t>("first_index")
        .Arg<ARG>().Ret<RET>().Ret<RET>());
XLA_FFI_DEFINE_HANDLER_SYMBOL(gpvae_krig_predict_ffi, KrigPredict,
    ffi::Ffi::Bind().Ctx<Stream>().Ctx<ffi::ScratchAllocator>().Arg<ARG>().Arg<ARG>().Arg<ARG>().Arg<ARG>().Arg<ARG>()
        .Arg<ARG>().Arg<ARG>().Attr<double>("jitter").Ret<RET>().Ret<RET>().Ret<RET>().Ret<ffi::Buffer<ffi::S32>>());
XLA_FFI_DEFINE_HANDLER_SYMBOL(gpvae_mvn_logp_ffi, MvnLogp,
    ffi::Ffi::Bind().Ctx<Stream>().Ctx<ffi::ScratchAllocator>().Arg<ARG>().Arg<ARG>().Arg<ARG>().Arg<ARG>()
        .Attr<double>("jitter").Ret<RET>().Ret<RET>().Ret<RET>().Ret<RET>().Ret<ffi::Buffer<ffi::S32>>());
XLA_FFI_DEFINE_HANDLER_SYMBOL(gpvae_gp_potential_ffi, GpPotential,
    ffi::Ffi::Bind().Ctx<Stream>().Ctx<ffi::ScratchAllocator>().Arg<ARG>().Arg<ARG>().Arg<ARG>().Arg<ffi::Buffer<ffi::S64>>()
        .Attr<double>("jitter").Ret<RET>().Ret<RET>().Ret<ffi::Buffer<ffi::S32>>());
XLA_FFI_DEFINE_HANDLER_SYMBOL(gpvae_vae_elbo_ffi, VaeElbo,
    ffi::Ffi::Bind().Ctx<Stream>().Ctx<ffi::ScratchAllocator>().Arg<ARG>().Arg<ARG>().Arg<ARG>().Attr<int32_t>("h0")
        .Attr<int32_t>("h1").Attr<double>("obs_sigma").Attr<bool>("with_grads").Ret<RET>().Ret<RET>());
XLA_FFI_DEFINE_HANDLER_SYMBOL(gpvae_encoder_fwd_ffi, EncoderFwd,
    ffi::Ffi::Bind().Ctx<Stream>().Ctx<ffi::ScratchAllocator>().Arg<ARG>().Arg<ARG>().Arg<ARG>().Attr<int32_t>("h0")
        .Attr<int32_t>("h1").Ret<RET>().Ret<RET>().Ret<RET>());
XLA_FFI_DEFINE_HANDLER_SYMBOL(gpvae_adam_ffi, AdamStep,
    ffi::Ffi::Bind().Ctx<Stream>().Arg<ARG>().Arg<ARG>().Arg<ARG>().Arg<ARG>().Attr<int64_t>("t").Attr<double>("lr")
        .Attr<double>("b1").Attr<double>("b2").Attr<double>("eps").Ret<RET>().Ret<RET>().Ret<RET>());
XLA_FFI_DEFINE_HANDLER_SYMBOL(gpvae_decoder_fwd_ffi, DecoderFwd,
    ffi::Ffi::Bind().Ctx<Stream>().Ctx<ffi::ScratchAllocator>().Arg<ARG>().Arg<ARG>().Attr<int32_t>("h0")
        .Attr<int32_t>("h1").Ret<RET>());
XLA_FFI_DEFINE_HANDLER_SYMBOL(gpvae_decoder_vjp_ffi, DecoderVjp,
    ffi::Ffi::Bind().Ctx<Stream>().Ctx<ffi::ScratchAllocator>().Arg<ARG>().Arg<ARG>().Arg<ARG>().Attr<int32_t>("h0")
        .Attr<int32_t>("h1").Ret<RET>());
XLA_FFI_DEFINE_HANDLER_SYMBOL(gpvae_vae_potential_ffi, VaePotential,
    ffi::Ffi::Bind().Ctx<Stream>().Arg<ARG>().Arg<ARG>().Arg<ARG>().Arg<ffi::Buffer<ffi::S64>>().Attr<int64_t>("n")
        .Attr<int32_t>("h0").Attr<int32_t>("h1").Attr<double>("noise_scale").Ret<RET>().Ret<RET>());
