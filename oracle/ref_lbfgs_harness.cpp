/*
 * ref_lbfgs_harness.cpp — TEST INFRASTRUCTURE ONLY.
 *
 * Thin extern "C" shim around the VERBATIM reference solver
 *   /root/reference/include/trajectory_planner/solver/lbfgs.hpp
 * which is header-only and depends on libc/libm only.  The header is included from where it
 * lies (the Makefile passes -I/root/reference/include); no reference source is copied here.
 * Output: oracle/_ref/libref_lbfgs.so (git-ignored, travels to the GPU box via gpurun).
 *
 * It lets tests pin oracle/vigo_oracle.c's L-BFGS restatement bit-for-bit against the
 * reference, and lets tests/golden/make_golden.py emit traces produced by the reference.
 */
#include <trajectory_planner/solver/lbfgs.hpp>

extern "C" {

typedef double (*ref_eval_fn)(void* ctx, const double* x, double* g, int n);
typedef void (*ref_trace_fn)(void* tctx, const double* x, const double* g, double fx, double step, int n);

struct ref_shim {
    ref_eval_fn eval;
    void* ctx;
    ref_trace_fn trace;
    void* tctx;
    int evals;
};

static double shim_eval(void* inst, const double* x, double* g, const int n) {
    ref_shim* s = static_cast<ref_shim*>(inst);
    ++s->evals;
    return s->eval(s->ctx, x, g, n);
}

static int shim_progress(void* inst, const double* x, const double* g, const double fx,
                         const double, const double, const double step, int n, int, int) {
    ref_shim* s = static_cast<ref_shim*>(inst);
    if (s->trace) s->trace(s->tctx, x, g, fx, step, n);
    return 0;
}

/* params: {mem_size, max_iterations, max_linesearch, past} and
 * {g_epsilon, delta, min_step, max_step, f_dec_coeff, s_curv_coeff, xtol} */
int ref_lbfgs_optimize(int n, double* x, double* fx, ref_eval_fn eval, void* ctx,
                       const int* iparams, const double* dparams, int* out_evals,
                       ref_trace_fn trace, void* tctx) {
    lbfgs::lbfgs_parameter_t p;
    lbfgs::lbfgs_load_default_parameters(&p);
    p.mem_size = iparams[0];
    p.max_iterations = iparams[1];
    p.max_linesearch = iparams[2];
    p.past = iparams[3];
    p.g_epsilon = dparams[0];
    p.delta = dparams[1];
    p.min_step = dparams[2];
    p.max_step = dparams[3];
    p.f_dec_coeff = dparams[4];
    p.s_curv_coeff = dparams[5];
    p.xtol = dparams[6];
    ref_shim s = {eval, ctx, trace, tctx, 0};
    int ret = lbfgs::lbfgs_optimize(n, x, fx, shim_eval, NULL, trace ? shim_progress : NULL, &s, &p);
    if (out_evals) *out_evals = s.evals;
    return ret;
}

}  // extern "C"
