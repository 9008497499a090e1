// api_eval.h -- primitive evaluation hooks (jk_host_eval / jk_dev_eval) used by the tests
// (part of the one translation unit jk_api.hip; see the include list there)
#pragma once

namespace jk {

// ---- primitive evaluation hooks ---------------------------------------------------------------
static double g_eval_shape = 16.0, g_eval_scale = 25.0;

struct EvalRng { jk_pcg64 e; JK_HD uint64_t operator()() { return jk_pcg_next(e); } };

JK_HD void eval_one(int what, const uint64_t* in, uint64_t i, uint64_t aux, uint64_t* out, jk_gamma_param gp) {
    switch (what) {
        case JK_OP_PCG_STREAM: {
            uint32_t w[8];
            for (int k = 0; k < 8; k++) w[k] = (uint32_t)in[i * 8 + k];
            jk_pcg64 e = jk_pcg_seed(w);
            for (uint64_t k = 0; k < aux; k++) out[i * aux + k] = jk_pcg_next(e);
            break;
        }
        case JK_OP_RUNIF_INDEX: out[i] = jk_runif_index(in[i], aux); break;
        case JK_OP_RUNIF_DOUBLE: out[i] = jk_d2u(jk_runif_double(in[i])); break;
        case JK_OP_CANONICAL: out[i] = jk_d2u(jk_canonical(in[i])); break;
        case JK_OP_N_QUAL:                 // on the device: the kernels' form (common path + exact routine behind a rare branch)
#if defined(__HIP_DEVICE_COMPILE__)
            out[i] = n_qual32(in[i]);
#else
            out[i] = jk_n_qual(in[i]);
#endif
            break;
        case JK_OP_LT_HALF: out[i] = jk_runif_lt_half(in[i]) ? 1 : 0; break;
        case JK_OP_FRAG_START: out[i] = jk_frag_start(in[i], aux); break;
        case JK_OP_LOG: out[i] = jk_d2u(jk_log(jk_u2d(in[i]))); break;
        case JK_OP_SQRT: out[i] = jk_d2u(jk_sqrt(jk_u2d(in[i]))); break;
        case JK_OP_GAMMA_STREAM: {
            uint32_t w[8];
            for (int k = 0; k < 8; k++) w[k] = (uint32_t)in[i * 8 + k];
            EvalRng r; r.e = jk_pcg_seed(w);
            jk_gamma_state st; st.saved = 0; st.saved_available = 0; st.fail = 0;
            for (uint64_t k = 0; k < aux; k++) out[i * aux + k] = jk_d2u(jk_gamma(gp, st, r));
            break;
        }
        case JK_OP_EXP: { double r = 0; bool ok = jk_exp(jk_u2d(in[i]), &r); out[i] = ok ? jk_d2u(r) : ~0ULL; break; }
        case JK_OP_POW: { bool ok = true; double r = jk_pow(jk_u2d(in[2 * i]), jk_u2d(in[2 * i + 1]), &ok); out[i] = ok ? jk_d2u(r) : ~0ULL; break; }
        case JK_OP_LOG10: out[i] = jk_d2u(jk_log10(jk_u2d(in[i]))); break;
        case JK_OP_QNORM: out[i] = jk_d2u(jk_qnorm(jk_u2d(in[i]))); break;
        case JK_OP_RUNIF_AB: {
            jk_x87 c; c.m = in[4 * i + 2]; c.e = (int32_t)(int64_t)in[4 * i + 3];
            out[i] = jk_d2u(jk_runif_ab(in[4 * i], jk_x87_from_double(jk_u2d(in[4 * i + 1])), c));
            break;
        }
        case JK_OP_RUNIF_INDEX32:          // the kernels' 32-bit form of RUNIF_INDEX (n < 2^32), device only
#if defined(__HIP_DEVICE_COMPILE__)
            out[i] = runif_index32(in[i], (uint32_t)aux);
#else
            out[i] = jk_runif_index(in[i], aux);
#endif
            break;
        case JK_OP_ALIAS_INDEX32:          // the quality step's index routine (table sizes <= 255), device only
#if defined(__HIP_DEVICE_COMPILE__)
            out[i] = alias_index32(in[i], (uint32_t)aux);
#else
            out[i] = jk_runif_index(in[i], aux);
#endif
            break;
        default: break;
    }
}

__global__ void eval_kernel(int what, const uint64_t* in, uint64_t n, uint64_t aux, uint64_t* out, jk_gamma_param gp) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) eval_one(what, in, i, aux, out, gp);
}

static jk_gamma_param eval_gamma_param() {
    const jk_gamma_param gp = jk_gamma_make(g_eval_shape, g_eval_scale);
    return gp;
}

static void eval_sizes(int what, uint64_t n, uint64_t aux, uint64_t* n_in, uint64_t* n_out) {
    const bool stream = (what == JK_OP_PCG_STREAM || what == JK_OP_GAMMA_STREAM);
    *n_in = stream ? n * 8 : (what == JK_OP_POW ? n * 2 : (what == JK_OP_RUNIF_AB ? n * 4 : n));
    *n_out = stream ? n * aux : n;
}

}  // namespace jk
