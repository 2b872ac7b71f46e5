// TEST INFRASTRUCTURE ONLY (oracle/_ref build).
//
// Stand-in for robotology/osqp-eigen's OsqpEigen::Solver: the call surface the reference uses
// (S/A1RobotControl.cpp:416-439, 522-555; S/test/test_mpc.cpp:131-151), backed by the oracle's restatement of the
// OSQP 0.6 algorithm (orc_osqp_solve, oracle/a1mpc_oracle.c).  Real OSQP / osqp-eigen are not on this machine, so
// the SOLVE half of anything built on this header is still "parity unpinned" -- what it makes checkable is that
// the reference's own sources hand OSQP the QP data, the warm-start sequence and the settings we think they do.
#pragma once
#include <memory>
#include <vector>
#include "mini_eigen.hpp"
#include "../../a1mpc_oracle_api.h"

namespace OsqpEigen {
const double INFTY = 1e30;   // OSQP's OSQP_INFTY

// what the last solve() of any Solver saw and produced (read by oracle/ref_harness.cpp)
struct ShimRecord {
    int n = 0, m = 0, solves = 0, inits = 0, reinits = 0;   // reinits: updateHessianMatrix calls that found another sparsity pattern
    std::vector<double> P, q, A, l, u, x, y;   // P n*n row-major (full symmetric), A m*n row-major
    orc_info info{};
};
inline ShimRecord &shim_last() { static ShimRecord r; return r; }
// settings every Solver starts from (tests may change it before constructing the controller); default = OSQP defaults
inline orc_settings &shim_base_settings() { static orc_settings s = [] { orc_settings t; orc_default_settings(&t); return t; }(); return s; }

class Settings {
  public:
    orc_settings st = shim_base_settings();
    void setVerbosity(bool) {}
    void setWarmStart(bool w) { st.warm_start = w ? 1 : 0; }
    void setAbsoluteTolerance(double v) { st.eps_abs = v; }
    void setRelativeTolerance(double v) { st.eps_rel = v; }
    void setMaxIteration(int v) { st.max_iter = v; }
};

class Data {
  public:
    int n = 0, m = 0;
    std::vector<double> P, q, A, l, u;
    bool hasP = false, hasq = false, hasA = false, hasl = false, hasu = false;
    void setNumberOfVariables(int v) { n = v; }
    void setNumberOfConstraints(int v) { m = v; }
    bool setHessianMatrix(const Eigen::SparseMatrix<double> &H) {
        if (H.rows() != n || H.cols() != n) return false;
        P.assign((size_t)n * n, 0.0);
        // OSQP is given the upper triangle (osqp-eigen extracts it); the stored matrix is symmetric, keep it full
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) P[(size_t)i * n + j] = i <= j ? H.coeff(i, j) : H.coeff(j, i);
        hasP = true; return true;
    }
    bool setLinearConstraintsMatrix(const Eigen::SparseMatrix<double> &M) {
        if (M.rows() != m || M.cols() != n) return false;
        A.assign((size_t)m * n, 0.0);
        for (int i = 0; i < m; ++i) for (int j = 0; j < n; ++j) if (M.stored(i, j)) A[(size_t)i * n + j] = M.coeff(i, j);
        hasA = true; return true;
    }
    bool setGradient(const Eigen::Dyn<double> &g) { if (g.size() != n) return false; q.assign(g.data(), g.data() + n); hasq = true; return true; }
    bool setLowerBound(const Eigen::Dyn<double> &v) { if (v.size() != m) return false; l.assign(v.data(), v.data() + m); hasl = true; return true; }
    bool setUpperBound(const Eigen::Dyn<double> &v) { if (v.size() != m) return false; u.assign(v.data(), v.data() + m); hasu = true; return true; }
};

class Solver {
    std::unique_ptr<Settings> s_{new Settings};
    std::unique_ptr<Data> d_{new Data};
    bool init_ = false;
    std::vector<double> x_, y_; double rho_ = 0;   // last (x, y) and rho (cold / one-off solves)
    std::vector<double> carry_;                    // the persistent OSQP workspace of a warm-started solver between ticks: update*() + solve() follow OSQP's UPDATE path
                                                   // (orc_osqp_solve_update: osqp_update_P re-equilibrating with the previous gradient, carried scaled iterates)
    std::vector<unsigned char> pattern_;           // non-zero flags of the upper triangle of the Hessian the workspace was set up / last updated with
    bool pattern_changed_ = false;                 // updateHessianMatrix found another pattern: the next solve() re-initialises (osqp-eigen 0.6.3, see updateHessianMatrix)
    static std::vector<unsigned char> pattern_of(const Eigen::SparseMatrix<double> &H) {
        // what osqp-eigen compares: the triplets of the upper triangle of H (H comes from dense.sparseView(): exact zeros are not stored)
        std::vector<unsigned char> p; const int n = H.rows();
        for (int i = 0; i < n; ++i) for (int j = i; j < n; ++j) p.push_back(H.stored(i, j) && H.coeff(i, j) != 0.0 ? 1 : 0);
        return p;
    }
    Eigen::VectorXd sol_;
  public:
    const std::unique_ptr<Settings> &settings() const { return s_; }
    const std::unique_ptr<Data> &data() const { return d_; }
    bool isInitialized() const { return init_; }
    bool initSolver() {
        if (!(d_->hasP && d_->hasq && d_->hasA && d_->hasl && d_->hasu)) return false;
        init_ = true; x_.assign((size_t)d_->n, 0.0); y_.assign((size_t)d_->m, 0.0); rho_ = 0; shim_last().inits++;
        carry_.assign((size_t)(2 + 2 * d_->n + 4 * d_->m), 0.0);
        pattern_.clear(); pattern_changed_ = false;
        {   // the pattern the workspace is set up with (Data keeps the full symmetric matrix)
            const int n = d_->n;
            for (int i = 0; i < n; ++i) for (int j = i; j < n; ++j) pattern_.push_back(d_->P[(size_t)i * n + j] != 0.0 ? 1 : 0);
        }
        return true;
    }
    void clearSolver() { init_ = false; }
    // osqp-eigen 0.6.3 Solver::updateHessianMatrix: the new upper-triangular triplets are compared with those of the workspace's P.  Same pattern: osqp_update_P
    // (the UPDATE path).  Another pattern -- the reference's dense B_qp'QB_qp gained or lost exact zeros (S/ConvexMpc.cpp:211 sparseView) --: the primal / dual
    // workspace iterates are read, the solver is cleared and initialised again with the new Hessian, and the iterates go back in through osqp_warm_start_x / _y
    // (orc_osqp_solve_update_ex with pattern_changed = 1 restates exactly that on the carried workspace).
    bool updateHessianMatrix(const Eigen::SparseMatrix<double> &H) {
        if (!init_ || !d_->setHessianMatrix(H)) return false;
        std::vector<unsigned char> p = pattern_of(H);
        if (p != pattern_) { pattern_changed_ = true; pattern_ = p; shim_last().reinits++; }
        return true;
    }
    bool updateGradient(const Eigen::Dyn<double> &g) { return init_ && d_->setGradient(g); }
    bool updateLowerBound(const Eigen::Dyn<double> &v) { return init_ && d_->setLowerBound(v); }
    bool updateUpperBound(const Eigen::Dyn<double> &v) { return init_ && d_->setUpperBound(v); }
    bool updateBounds(const Eigen::Dyn<double> &l, const Eigen::Dyn<double> &u) { return updateLowerBound(l) && updateUpperBound(u); }
    bool solve() {
        if (!init_) return false;
        const int n = d_->n, m = d_->m;
        std::vector<int32_t> rp((size_t)m + 1), ci; std::vector<double> av;
        for (int i = 0; i < m; ++i) { rp[(size_t)i] = (int32_t)ci.size(); for (int j = 0; j < n; ++j) if (d_->A[(size_t)i * n + j] != 0.0) { ci.push_back(j); av.push_back(d_->A[(size_t)i * n + j]); } }
        rp[(size_t)m] = (int32_t)ci.size();
        const orc_settings &st = s_->st;
        if (!st.warm_start) { x_.assign((size_t)n, 0.0); y_.assign((size_t)m, 0.0); rho_ = 0; }
        ShimRecord &rec = shim_last();
        rec.n = n; rec.m = m; rec.P = d_->P; rec.q = d_->q; rec.A = d_->A; rec.l = d_->l; rec.u = d_->u;
        int rc;
        if (st.warm_start) {   // the reference's MPC solver: initSolver once, then updateHessianMatrix / updateGradient / update*Bound + solve every tick
            x_.assign((size_t)n, 0.0); y_.assign((size_t)m, 0.0);
            rc = orc_osqp_solve_update_ex(n, m, d_->P.data(), d_->q.data(), rp.data(), ci.data(), av.data(), d_->l.data(), d_->u.data(), &st,
                                          x_.data(), y_.data(), carry_.data(), pattern_changed_ ? 1 : 0, &rec.info);
            rec.info.reinit = pattern_changed_ ? 1 : 0;
            pattern_changed_ = false;
        } else {
            rc = orc_osqp_solve(n, m, d_->P.data(), d_->q.data(), rp.data(), ci.data(), av.data(), d_->l.data(), d_->u.data(), &st,
                                x_.data(), y_.data(), &rho_, &rec.info);
        }
        rec.x = x_; rec.y = y_; rec.solves++;
        sol_.resize(n); for (int j = 0; j < n; ++j) sol_(j) = x_[(size_t)j];
        // OSQP cold-starts its iterates after a failed solve (osqp_solve -> store_solution -> cold_start); like orc_mpc_solve the next
        // tick is a full cold start (rho back to settings->rho as well)
        bool failed = false; for (int j = 0; j < n; ++j) failed |= std::isnan(x_[(size_t)j]);
        if (failed) { x_.assign((size_t)n, 0.0); y_.assign((size_t)m, 0.0); rho_ = 0; }
        return rc == 0;
    }
    Eigen::VectorXd getSolution() const { return sol_; }
};
}  // namespace OsqpEigen
