// C ABI of include/mgadmm.h: solver handles and the entry points that forward to the engines.
#include "common.h"

// ------------------------------------------------------------------------------------ C ABI
extern "C" {

int mgadmm_solver_create(mgadmm_graph* g, const mgadmm_params* p, int32_t max_batch, mgadmm_solver** out) {
    MG_REQUIRE(g && p && out, "solver_create: null argument");
    MG_REQUIRE(max_batch >= 1, "solver_create: max_batch must be >= 1");
    MG_REQUIRE(p->dtype == MGADMM_F32 || p->dtype == MGADMM_F64, "solver_create: bad dtype %d", p->dtype);
    MG_REQUIRE(p->t_in >= 1 && p->t_in <= g->T, "solver_create: t_in %d outside [1, T=%d]", p->t_in, g->T);
    MG_REQUIRE(p->ablation >= 0 && p->ablation <= 3, "solver_create: ablation should be one of None, DGTV, DGLR, UT");
    MG_REQUIRE(p->max_cg_iter >= 1 && p->max_admm_iter >= 1, "solver_create: iteration limits must be >= 1");
    MG_REQUIRE(p->cg_convergence == MGADMM_CG_PER_SAMPLE || p->cg_convergence == MGADMM_CG_BATCH_MAX,
               "solver_create: cg_convergence should be per_sample (0) or batch_max (1), got %d", p->cg_convergence);
    if (p->path == MGADMM_PATH_LDS && p->cg_convergence == MGADMM_CG_BATCH_MAX) {
        mg_set_error("solver_create: the LDS-resident path implements per-sample CG convergence only (batch_max: streaming path)");
        return MGADMM_ERR_UNSUPPORTED;
    }
    mgadmm_solver* s = new mgadmm_solver();
    s->g = g;
    s->p = *p;
    s->Bmax = max_batch;
    s->eng = p->dtype == MGADMM_F32 ? mg_make_engine_f32(s) : mg_make_engine_f64(s);
    int rc = s->eng->init();
    if (rc != MGADMM_OK) {
        delete s->eng;
        delete s;
        return rc;
    }
    *out = s;
    return MGADMM_OK;
}

int mgadmm_solver_destroy(mgadmm_solver* s) {
    if (!s) return MGADMM_OK;
    delete s->eng;
    delete s;
    return MGADMM_OK;
}

int mgadmm_solver_set_params(mgadmm_solver* s, const mgadmm_params* p) {
    MG_REQUIRE(s && p, "set_params: null argument");
    return s->eng->set_params(*p);
}

int64_t mgadmm_solver_workspace_bytes(const mgadmm_solver* s) { return s ? s->eng->workspace_bytes() : 0; }
int mgadmm_solver_path(const mgadmm_solver* s, int32_t B) { return s ? s->eng->path_for(B) : MGADMM_ERR_INVALID; }

int mgadmm_solver_query(const mgadmm_solver* s, int32_t what, int64_t* out) {
    MG_REQUIRE(s && out, "solver_query: null argument");
    return s->eng->query(what, out);
}

int mgadmm_apply(mgadmm_solver* s, int32_t op, const void* x, void* y, int32_t B, void* stream) {
    MG_REQUIRE(s, "apply: null solver");
    return s->eng->apply(op, x, y, B, (hipStream_t)stream);
}
int mgadmm_lhs(mgadmm_solver* s, int32_t which, const void* x, const void* mask, void* y, int32_t B, void* stream) {
    MG_REQUIRE(s, "lhs: null solver");
    return s->eng->lhs(which, x, mask, y, B, (hipStream_t)stream);
}
int mgadmm_phi_direct(mgadmm_solver* s, const void* x, const void* gamma, void* phi, int32_t B, void* stream) {
    MG_REQUIRE(s, "phi_direct: null solver");
    return s->eng->phi_direct(x, gamma, phi, B, (hipStream_t)stream);
}
int mgadmm_initial_guess(mgadmm_solver* s, const void* y, void* x, int32_t B, void* stream) {
    MG_REQUIRE(s, "initial_guess: null solver");
    return s->eng->initial_guess(y, x, B, (hipStream_t)stream);
}
int mgadmm_initial_interpolation(mgadmm_solver* s, const void* y, const void* mask, int32_t mask_is_f32, void* x,
                                 int32_t B, void* stream) {
    MG_REQUIRE(s, "initial_interpolation: null solver");
    return s->eng->initial_interpolation(y, mask, mask_is_f32, x, B, (hipStream_t)stream);
}
int mgadmm_cg(mgadmm_solver* s, int32_t which, const void* rhs, const void* x0, const void* mask, void* x, int32_t* iters,
              double* alpha, double* beta, int32_t B, void* stream) {
    MG_REQUIRE(s, "cg: null solver");
    return s->eng->cg(which, rhs, x0, mask, x, iters, alpha, beta, B, (hipStream_t)stream);
}
int mgadmm_solve(mgadmm_solver* s, const void* y, const void* mask, int32_t mask_is_f32, int32_t B, void* x_out,
                 const mgadmm_state* state_out, mgadmm_history* hist, void* stream) {
    MG_REQUIRE(s, "solve: null solver");
    return s->eng->solve(y, mask, mask_is_f32, B, nullptr, nullptr, x_out, state_out, hist, (hipStream_t)stream);
}
int mgadmm_solve_from(mgadmm_solver* s, const void* y, const void* mask, int32_t mask_is_f32, int32_t B, const void* x0,
                      const mgadmm_state* state_in, void* x_out, const mgadmm_state* state_out, mgadmm_history* hist,
                      void* stream) {
    MG_REQUIRE(s, "solve_from: null solver");
    MG_REQUIRE(x0 && state_in, "solve_from: x0 and state_in are required (use mgadmm_solve for a cold start)");
    return s->eng->solve(y, mask, mask_is_f32, B, x0, state_in, x_out, state_out, hist, (hipStream_t)stream);
}
int mgadmm_two_loops(mgadmm_solver* s, const void* y, const void* mask, int32_t mask_is_f32, int32_t B, void* x_out,
                     const mgadmm_state* state_out, mgadmm_history* hist, void* stream) {
    MG_REQUIRE(s, "two_loops: null solver");
    return s->eng->two_loops(y, mask, mask_is_f32, B, x_out, state_out, hist, (hipStream_t)stream);
}
int mgadmm_prof_begin(mgadmm_solver* s) {
    MG_REQUIRE(s, "prof_begin: null solver");
    return s->eng->prof_begin();
}
int mgadmm_prof_end(mgadmm_solver* s, int64_t* counts, double* total_ms, double* bytes) {
    MG_REQUIRE(s && counts && total_ms && bytes, "prof_end: null argument");
    return s->eng->prof_end(counts, total_ms, bytes);
}

}  // extern "C"
