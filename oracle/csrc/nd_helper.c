/* Extend-add of a child's update matrix into its parent's front for oracle/nd_lu.py (TEST INFRASTRUCTURE, CPU only).
 *
 * The multifrontal method (the algorithm behind the reference's `pc_factor_mat_solver_type mumps`,
 * /root/reference/examples/01_obstacle_problem/obstacle_pg.py:129-131) adds the Schur complement U (bc x bc, column-major) of every
 * child into the parent's front at the positions `pos` of the child's border dofs.  The parent's front is stored as four
 * column-major blocks F11 (p x p), F12 (p x b), F21 (b x p), F22 (b x b); pos[] is ascending, its first k entries are < p.
 * numpy's fancy indexing does this at ~20 M entries/s; this loop is memory-bound.
 *
 * Build: gcc -O3 -shared -fPIC -o oracle/_build/libndhelper.so oracle/csrc/nd_helper.c   (done by __graft_entry__.build()) */
#include <stdint.h>

void nd_extend_add(double* F11, double* F12, double* F21, double* F22, int64_t p, int64_t b, const double* U, int64_t bc,
                   const int64_t* pos, int64_t k)
{
    for (int64_t j = 0; j < bc; ++j) {
        const double* u = U + j * bc;
        const int64_t cj = pos[j];
        double *top, *bot; /* destination column: rows < p, rows >= p */
        if (cj < p) { top = F11 + cj * p; bot = F21 + cj * b; }
        else        { top = F12 + (cj - p) * p; bot = F22 + (cj - p) * b; }
        for (int64_t i = 0; i < k; ++i) top[pos[i]] += u[i];
        for (int64_t i = k; i < bc; ++i) bot[pos[i] - p] += u[i];
    }
}

/* buf[dst[i]] = data[src[i]]: the original matrix entries of one front */
void nd_scatter(double* buf, const int64_t* dst, const double* data, const int64_t* src, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) buf[dst[i]] = data[src[i]];
}
