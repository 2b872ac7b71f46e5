// mini_eigen.hpp -- TEST INFRASTRUCTURE ONLY (oracle/_ref build).
//
// A small, eager (no expression templates) stand-in for the subset of the Eigen 3 API that the reference's
// hot-path sources use, so that those sources can be compiled VERBATIM from /root/reference (Eigen itself is
// not installed on this machine and there is no network).  It is our own code, not a copy of Eigen: every
// operation evaluates immediately into a heap-backed column-major matrix.  What that pins and what it does
// not: the reference's *source logic* (which blocks are multiplied, index arithmetic, the order of the
// statements) runs exactly as written; the *rounding of the primitives* is this file's (plain ascending-k
// dot products, cofactor 3x3 inverse, unblocked partial-pivot LU), not Eigen's vectorised kernels.
// The product library never includes this file.
#pragma once
#include <cassert>
#include <cmath>
#include <cstdlib>
#include <stdlib.h>   // real Eigen pulls in <stdlib.h> on x86 (emmintrin.h -> mm_malloc.h); with libstdc++ that puts std::abs(double)
                      // into the global namespace, which the unqualified abs() of S/utils/Utils.cpp:57 depends on (int abs() otherwise)
#include <cstring>
#include <iostream>
#include <limits>
#include <stdexcept>
#include <vector>

namespace Eigen {

typedef long Index;
const int Dynamic = -1;
enum { ComputeFullU = 1, ComputeFullV = 2, ComputeThinU = 4, ComputeThinV = 8 };

template <class T> class Dyn;
template <class T> class Ref;
template <class T, int R, int C> class Matrix;
template <class T, int N> class DiagonalMatrix;
template <class T> class SparseMatrix;
template <class T> struct PartialPivLU;
template <class T> struct GaussSolver;
template <class T> struct Arr;

inline void shim_check(bool ok, const char *what) {
    if (!ok) throw std::runtime_error(std::string("mini_eigen: ") + what);
}

// comma initialiser: row-major fill of a rows x cols window of a Dyn
template <class T> struct CommaInit {
    Dyn<T> *m; Index r0, c0, nr, nc, k;
    template <class S> CommaInit &operator,(const S &v);
};

// ---------------------------------------------------------------------------------------------------------
template <class T> class Dyn {
  public:
    std::vector<T> d; Index r = 0, c = 0;
    Dyn() {}
    Dyn(Index rows, Index cols, int) : d((size_t)(rows * cols), T(0)), r(rows), c(cols) {}  // tagged: never a converting ctor
    Index rows() const { return r; }
    Index cols() const { return c; }
    Index size() const { return r * c; }
    T *data() { return d.data(); }
    const T *data() const { return d.data(); }
    void resize(Index n) { if (c == 1 || r == 0) { r = n; c = 1; } else { shim_check(r == 1, "resize(n) on a matrix"); c = n; } d.assign((size_t)(r * c), T(0)); }
    void resize(Index rows, Index cols) { r = rows; c = cols; d.assign((size_t)(r * c), T(0)); }
    T &operator()(Index i, Index j) { return d[(size_t)(j * r + i)]; }
    const T &operator()(Index i, Index j) const { return d[(size_t)(j * r + i)]; }
    T &operator()(Index i) { return d[(size_t)i]; }
    const T &operator()(Index i) const { return d[(size_t)i]; }
    T &operator[](Index i) { return d[(size_t)i]; }
    const T &operator[](Index i) const { return d[(size_t)i]; }
    void setZero() { for (auto &v : d) v = T(0); }
    void setOnes() { for (auto &v : d) v = T(1); }
    void setIdentity() { setZero(); for (Index i = 0; i < (r < c ? r : c); ++i) (*this)(i, i) = T(1); }
    void setConstant(T v) { for (auto &e : d) e = v; }

    Ref<T> block(Index i, Index j, Index nr, Index nc) { return Ref<T>(this, i, j, nr, nc); }
    Dyn<T> block(Index i, Index j, Index nr, Index nc) const { Dyn<T> o(nr, nc, 0); for (Index b = 0; b < nc; ++b) for (Index a = 0; a < nr; ++a) o(a, b) = (*this)(i + a, j + b); return o; }
    template <int NR, int NC> Ref<T> block(Index i, Index j) { return Ref<T>(this, i, j, NR, NC); }
    template <int NR, int NC> Dyn<T> block(Index i, Index j) const { return block(i, j, NR, NC); }
    Ref<T> segment(Index i, Index n) { return c == 1 ? Ref<T>(this, i, 0, n, 1) : Ref<T>(this, 0, i, 1, n); }
    Dyn<T> segment(Index i, Index n) const { return c == 1 ? block(i, 0, n, 1) : block(0, i, 1, n); }
    template <int N> Ref<T> segment(Index i) { return segment(i, N); }
    template <int N> Dyn<T> segment(Index i) const { return segment(i, N); }
    Ref<T> col(Index j) { return Ref<T>(this, 0, j, r, 1); }
    Ref<T> row(Index i) { return Ref<T>(this, i, 0, 1, c); }

    Dyn<T> transpose() const { Dyn<T> o(c, r, 0); for (Index j = 0; j < c; ++j) for (Index i = 0; i < r; ++i) o(j, i) = (*this)(i, j); return o; }
    Dyn<T> adjoint() const { return transpose(); }
    Dyn<T> cwiseProduct(const Dyn<T> &b) const { shim_check(r == b.r && c == b.c, "cwiseProduct size"); Dyn<T> o(r, c, 0); for (size_t k = 0; k < d.size(); ++k) o.d[k] = d[k] * b.d[k]; return o; }
    T norm() const { T s = 0; for (auto v : d) s += v * v; return std::sqrt(s); }
    T determinant() const;
    Dyn<T> inverse() const;
    Dyn<T> eval() const { return *this; }
    Dyn<T> &matrix() { return *this; }
    Arr<T> array() const;
    DiagonalMatrix<T, Dynamic> asDiagonal() const;
    SparseMatrix<T> sparseView() const;
    PartialPivLU<T> lu() const;
    GaussSolver<T> fullPivHouseholderQr() const;
    Dyn<T> eulerAngles(int a0, int a1, int a2) const;

    CommaInit<T> operator<<(const T &v) { CommaInit<T> ci{this, 0, 0, r, c, 0}; ci, v; return ci; }
    Dyn<T> &operator+=(const Dyn<T> &b) { shim_check(r == b.r && c == b.c, "+= size"); for (size_t k = 0; k < d.size(); ++k) d[k] += b.d[k]; return *this; }
    Dyn<T> &operator-=(const Dyn<T> &b) { shim_check(r == b.r && c == b.c, "-= size"); for (size_t k = 0; k < d.size(); ++k) d[k] -= b.d[k]; return *this; }
    Dyn<T> &operator*=(T s) { for (auto &v : d) v *= s; return *this; }
    Dyn<T> &operator/=(T s) { for (auto &v : d) v /= s; return *this; }
    template <int N> Dyn<T> &operator+=(const DiagonalMatrix<T, N> &D);
};

template <class T> template <class S> CommaInit<T> &CommaInit<T>::operator,(const S &v) {
    shim_check(k < nr * nc, "comma initialiser: too many coefficients");
    (*m)(r0 + k / nc, c0 + k % nc) = (T)v; ++k; return *this;
}

// ---------------------------------------------------------------------------------------------------------
template <class T> class Ref {
  public:
    Dyn<T> *m; Index r0, c0, nr, nc;
    Ref(Dyn<T> *m_, Index i, Index j, Index a, Index b) : m(m_), r0(i), c0(j), nr(a), nc(b) { shim_check(i >= 0 && j >= 0 && i + a <= m_->r && j + b <= m_->c, "block out of range"); }
    Ref(const Ref &) = default;
    operator Dyn<T>() const { Dyn<T> o(nr, nc, 0); for (Index b = 0; b < nc; ++b) for (Index a = 0; a < nr; ++a) o(a, b) = (*m)(r0 + a, c0 + b); return o; }
    Dyn<T> eval() const { return (Dyn<T>)*this; }
    Index rows() const { return nr; }
    Index cols() const { return nc; }
    Index size() const { return nr * nc; }
    Ref &operator=(const Dyn<T> &v) {
        if (v.r == nr && v.c == nc) { for (Index b = 0; b < nc; ++b) for (Index a = 0; a < nr; ++a) (*m)(r0 + a, c0 + b) = v(a, b); }
        else { shim_check((nr == 1 || nc == 1) && (v.r == 1 || v.c == 1) && v.size() == nr * nc, "block assignment size"); for (Index k = 0; k < nr * nc; ++k) (*this)(k) = v.d[(size_t)k]; }
        return *this;
    }
    Ref &operator=(const Ref &o) { Dyn<T> t = o; return (*this = t); }
    T &operator()(Index i, Index j) { return (*m)(r0 + i, c0 + j); }
    T operator()(Index i, Index j) const { return (*m)(r0 + i, c0 + j); }
    T &operator()(Index i) { return nc == 1 ? (*m)(r0 + i, c0) : (*m)(r0, c0 + i); }
    T operator()(Index i) const { return nc == 1 ? (*m)(r0 + i, c0) : (*m)(r0, c0 + i); }
    T &operator[](Index i) { return (*this)(i); }
    void setZero() { for (Index b = 0; b < nc; ++b) for (Index a = 0; a < nr; ++a) (*this)(a, b) = T(0); }
    void setOnes() { for (Index b = 0; b < nc; ++b) for (Index a = 0; a < nr; ++a) (*this)(a, b) = T(1); }
    void setIdentity() { for (Index b = 0; b < nc; ++b) for (Index a = 0; a < nr; ++a) (*this)(a, b) = a == b ? T(1) : T(0); }
    Dyn<T> transpose() const { return eval().transpose(); }
    Dyn<T> cwiseProduct(const Dyn<T> &b) const { return eval().cwiseProduct(b); }
    T norm() const { return eval().norm(); }
    T determinant() const { return eval().determinant(); }
    CommaInit<T> operator<<(const T &v) { CommaInit<T> ci{m, r0, c0, nr, nc, 0}; ci, v; return ci; }
    Ref &operator+=(const Dyn<T> &v) { Dyn<T> t = *this; t += v; return (*this = t); }
    Ref &operator-=(const Dyn<T> &v) { Dyn<T> t = *this; t -= v; return (*this = t); }
    Ref &operator/=(T s) { for (Index b = 0; b < nc; ++b) for (Index a = 0; a < nr; ++a) (*this)(a, b) /= s; return *this; }
    Ref &operator*=(T s) { for (Index b = 0; b < nc; ++b) for (Index a = 0; a < nr; ++a) (*this)(a, b) *= s; return *this; }
};

// ---------------------------------------------------------------------------------------------------------
template <class T, int R, int C> class Matrix : public Dyn<T> {
    void init_fixed() { this->r = R == Dynamic ? 0 : R; this->c = C == Dynamic ? (C == 1 ? 1 : 0) : C; if (C == 1) this->c = 1; this->d.assign((size_t)(this->r * this->c), T(0)); }
    void take(const Dyn<T> &v) {
        if ((R == Dynamic || R == v.r) && (C == Dynamic || C == v.c)) { this->r = v.r; this->c = v.c; this->d = v.d; return; }
        // vectors: automatic transposition as in Eigen
        shim_check((v.r == 1 || v.c == 1) && (R == 1 || C == 1), "matrix assignment: size mismatch");
        Index n = v.size();
        shim_check((R == 1 ? (C == Dynamic || C == n) : (R == Dynamic || R == n)), "vector assignment: size mismatch");
        if (C == 1) { this->r = n; this->c = 1; } else { this->r = 1; this->c = n; }
        this->d = v.d;
    }
  public:
    Matrix() { init_fixed(); }
    explicit Matrix(Index n) { init_fixed(); if (R == Dynamic || C == Dynamic) { if (C == 1) { this->r = n; this->c = 1; } else { this->r = 1; this->c = n; } this->d.assign((size_t)n, T(0)); } }
    Matrix(Index rows, Index cols) { init_fixed(); if (R == Dynamic && C == Dynamic) { this->r = rows; this->c = cols; this->d.assign((size_t)(rows * cols), T(0)); } else { shim_check(this->size() == 2, "2-coefficient ctor"); this->d[0] = (T)rows; this->d[1] = (T)cols; } }
    Matrix(T x, T y, T z) { init_fixed(); shim_check(this->size() == 3, "3-coefficient ctor"); this->d[0] = x; this->d[1] = y; this->d[2] = z; }
    Matrix(T x, T y, T z, T w) { init_fixed(); shim_check(this->size() == 4, "4-coefficient ctor"); this->d[0] = x; this->d[1] = y; this->d[2] = z; this->d[3] = w; }
    Matrix(const Dyn<T> &v) { init_fixed(); take(v); }
    Matrix(const Ref<T> &v) { init_fixed(); take((Dyn<T>)v); }
    Matrix(const Matrix &) = default;
    Matrix &operator=(const Matrix &) = default;
    Matrix &operator=(const Dyn<T> &v) { take(v); return *this; }
    Matrix &operator=(const Ref<T> &v) { take((Dyn<T>)v); return *this; }
    static Matrix Identity() { Matrix m; m.setIdentity(); return m; }
    static Matrix Zero() { Matrix m; return m; }
    static Matrix UnitX() { Matrix m; m.d[0] = 1; return m; }
    static Matrix UnitY() { Matrix m; m.d[1] = 1; return m; }
    static Matrix UnitZ() { Matrix m; m.d[2] = 1; return m; }
};

typedef Matrix<double, Dynamic, 1> VectorXd;
typedef Matrix<double, Dynamic, Dynamic> MatrixXd;
typedef Matrix<double, 3, 1> Vector3d;
typedef Matrix<double, 4, 1> Vector4d;
typedef Matrix<double, 3, 3> Matrix3d;

// ---------------------------------------------------------------------------------------------------------
// free operators (double only; blocks convert through Ref -> Dyn)
typedef Dyn<double> DynD;
inline DynD operator+(const DynD &a, const DynD &b) { shim_check(a.r == b.r && a.c == b.c, "+ size"); DynD o(a.r, a.c, 0); for (size_t k = 0; k < a.d.size(); ++k) o.d[k] = a.d[k] + b.d[k]; return o; }
inline DynD operator-(const DynD &a, const DynD &b) { shim_check(a.r == b.r && a.c == b.c, "- size"); DynD o(a.r, a.c, 0); for (size_t k = 0; k < a.d.size(); ++k) o.d[k] = a.d[k] - b.d[k]; return o; }
inline DynD operator-(const DynD &a) { DynD o(a.r, a.c, 0); for (size_t k = 0; k < a.d.size(); ++k) o.d[k] = -a.d[k]; return o; }
inline DynD operator*(const DynD &a, double s) { DynD o(a.r, a.c, 0); for (size_t k = 0; k < a.d.size(); ++k) o.d[k] = a.d[k] * s; return o; }
inline DynD operator*(double s, const DynD &a) { DynD o(a.r, a.c, 0); for (size_t k = 0; k < a.d.size(); ++k) o.d[k] = s * a.d[k]; return o; }
inline DynD operator/(const DynD &a, double s) { DynD o(a.r, a.c, 0); for (size_t k = 0; k < a.d.size(); ++k) o.d[k] = a.d[k] / s; return o; }
inline DynD operator*(const DynD &a, const DynD &b) {
    shim_check(a.c == b.r, "matrix product: inner sizes");
    DynD o(a.r, b.c, 0);
    for (Index j = 0; j < b.c; ++j) for (Index i = 0; i < a.r; ++i) { double s = 0; for (Index k = 0; k < a.c; ++k) s += a(i, k) * b(k, j); o(i, j) = s; }
    return o;
}
inline std::ostream &operator<<(std::ostream &os, const DynD &m) { for (Index i = 0; i < m.r; ++i) { for (Index j = 0; j < m.c; ++j) os << (j ? " " : "") << m(i, j); if (i + 1 < m.r) os << "\n"; } return os; }

// ---------------------------------------------------------------------------------------------------------
template <class T, int N> class DiagonalMatrix {
    Matrix<T, N, 1> dg;
  public:
    DiagonalMatrix() {}
    explicit DiagonalMatrix(const Dyn<T> &v) { dg = v; }
    Matrix<T, N, 1> &diagonal() { return dg; }
    const Matrix<T, N, 1> &diagonal() const { return dg; }
    Index rows() const { return dg.size(); }
};
template <int N> inline DynD operator*(const DynD &a, const DiagonalMatrix<double, N> &D) { shim_check(a.c == D.rows(), "M*diag size"); DynD o(a.r, a.c, 0); for (Index j = 0; j < a.c; ++j) for (Index i = 0; i < a.r; ++i) o(i, j) = a(i, j) * D.diagonal()(j); return o; }
template <int N> inline DynD operator*(const DiagonalMatrix<double, N> &D, const DynD &a) { shim_check(a.r == D.rows(), "diag*M size"); DynD o(a.r, a.c, 0); for (Index j = 0; j < a.c; ++j) for (Index i = 0; i < a.r; ++i) o(i, j) = D.diagonal()(i) * a(i, j); return o; }
template <class T> template <int N> Dyn<T> &Dyn<T>::operator+=(const DiagonalMatrix<T, N> &D) { shim_check(r == c && r == D.rows(), "+= diag size"); for (Index i = 0; i < r; ++i) (*this)(i, i) += D.diagonal()(i); return *this; }
template <class T> DiagonalMatrix<T, Dynamic> Dyn<T>::asDiagonal() const { return DiagonalMatrix<T, Dynamic>(*this); }

// ---------------------------------------------------------------------------------------------------------
// sparse matrix = dense values + structural-nonzero mask (sizes here are <= 200 x 120)
template <class T> class SparseMatrix {
  public:
    Dyn<T> dense; std::vector<char> mask;
    SparseMatrix() {}
    SparseMatrix(Index r, Index c) { resize(r, c); }
    void resize(Index r, Index c) { dense.resize(r, c); mask.assign((size_t)(r * c), 0); }
    Index rows() const { return dense.r; }
    Index cols() const { return dense.c; }
    Index nonZeros() const { Index n = 0; for (char m : mask) n += m; return n; }
    T &insert(Index i, Index j) { shim_check(!mask[(size_t)(j * dense.r + i)], "SparseMatrix::insert: entry exists"); mask[(size_t)(j * dense.r + i)] = 1; return dense(i, j); }
    T coeff(Index i, Index j) const { return dense(i, j); }
    bool stored(Index i, Index j) const { return mask[(size_t)(j * dense.r + i)] != 0; }
};
template <class T> SparseMatrix<T> Dyn<T>::sparseView() const {   // default reference = 0: drops exact zeros only
    SparseMatrix<T> s(r, c);
    for (Index j = 0; j < c; ++j) for (Index i = 0; i < r; ++i) if ((*this)(i, j) != T(0)) s.insert(i, j) = (*this)(i, j);
    return s;
}

// ---------------------------------------------------------------------------------------------------------
template <class T> T Dyn<T>::determinant() const {
    shim_check(r == c, "determinant: square");
    if (r == 2) return (*this)(0, 0) * (*this)(1, 1) - (*this)(1, 0) * (*this)(0, 1);
    if (r == 3) { const Dyn<T> &a = *this; return a(0, 0) * (a(1, 1) * a(2, 2) - a(1, 2) * a(2, 1)) - a(0, 1) * (a(1, 0) * a(2, 2) - a(1, 2) * a(2, 0)) + a(0, 2) * (a(1, 0) * a(2, 1) - a(1, 1) * a(2, 0)); }
    throw std::runtime_error("mini_eigen: determinant only for 2x2 / 3x3");
}
template <class T> Dyn<T> Dyn<T>::inverse() const {   // 3x3: cofactors / determinant
    shim_check(r == 3 && c == 3, "inverse only for 3x3");
    const Dyn<T> &a = *this; Dyn<T> o(3, 3, 0);
    T c00 = a(1, 1) * a(2, 2) - a(1, 2) * a(2, 1), c01 = a(1, 2) * a(2, 0) - a(1, 0) * a(2, 2), c02 = a(1, 0) * a(2, 1) - a(1, 1) * a(2, 0);
    T det = a(0, 0) * c00 + a(0, 1) * c01 + a(0, 2) * c02, id = T(1) / det;
    o(0, 0) = c00 * id; o(0, 1) = (a(0, 2) * a(2, 1) - a(0, 1) * a(2, 2)) * id; o(0, 2) = (a(0, 1) * a(1, 2) - a(0, 2) * a(1, 1)) * id;
    o(1, 0) = c01 * id; o(1, 1) = (a(0, 0) * a(2, 2) - a(0, 2) * a(2, 0)) * id; o(1, 2) = (a(0, 2) * a(1, 0) - a(0, 0) * a(1, 2)) * id;
    o(2, 0) = c02 * id; o(2, 1) = (a(0, 1) * a(2, 0) - a(0, 0) * a(2, 1)) * id; o(2, 2) = (a(0, 0) * a(1, 1) - a(0, 1) * a(1, 0)) * id;
    return o;
}

// unblocked partial-pivot LU (the published algorithm of Eigen's small-matrix path: first largest |pivot|, column
// divided by the pivot, rank-1 update of the trailing block), then permute / unit-lower / upper solves.
template <class T> struct PartialPivLU {
    Dyn<T> lu; std::vector<Index> piv;
    explicit PartialPivLU(const Dyn<T> &a) : lu(a), piv((size_t)a.r) {
        Index n = a.r; shim_check(a.r == a.c, "lu: square");
        for (Index k = 0; k < n; ++k) {
            Index p = k; T best = std::abs(lu(k, k));
            for (Index i = k + 1; i < n; ++i) if (std::abs(lu(i, k)) > best) { best = std::abs(lu(i, k)); p = i; }
            piv[(size_t)k] = p;
            if (best != T(0)) {
                if (p != k) for (Index j = 0; j < n; ++j) std::swap(lu(k, j), lu(p, j));
                for (Index i = k + 1; i < n; ++i) lu(i, k) /= lu(k, k);
            }
            for (Index j = k + 1; j < n; ++j) for (Index i = k + 1; i < n; ++i) lu(i, j) -= lu(i, k) * lu(k, j);
        }
    }
    Dyn<T> solve(const Dyn<T> &b) const {
        Index n = lu.r; Dyn<T> x = b;
        for (Index col = 0; col < x.c; ++col) {
            for (Index k = 0; k < n; ++k) std::swap(x(k, col), x(piv[(size_t)k], col));
            // row-oriented substitutions: the dot product of the already known part is formed first, then subtracted
            // (the order oracle/a1mpc_oracle.c orc_joint_torques uses; Eigen's own kernel order cannot be observed here)
            for (Index i = 1; i < n; ++i) { T s = lu(i, 0) * x(0, col); for (Index k = 1; k < i; ++k) s += lu(i, k) * x(k, col); x(i, col) -= s; }
            for (Index i = n - 1; i >= 0; --i) {
                if (i + 1 < n) { T s = lu(i, i + 1) * x(i + 1, col); for (Index k = i + 2; k < n; ++k) s += lu(i, k) * x(k, col); x(i, col) -= s; }
                x(i, col) /= lu(i, i);
            }
        }
        return x;
    }
};
template <class T> PartialPivLU<T> Dyn<T>::lu() const { return PartialPivLU<T>(*this); }

// stand-in for fullPivHouseholderQr().solve(): Gaussian elimination with partial pivoting (NOT Eigen's rank-revealing QR;
// results agree to solver accuracy, not bit for bit -- the tests that use it compare with a tolerance).
template <class T> struct GaussSolver {
    Dyn<T> a;
    explicit GaussSolver(const Dyn<T> &m) : a(m) {}
    Dyn<T> solve(const Dyn<T> &b) const { return PartialPivLU<T>(a).solve(b); }
};
template <class T> GaussSolver<T> Dyn<T>::fullPivHouseholderQr() const { return GaussSolver<T>(*this); }

// ---------------------------------------------------------------------------------------------------------
// array / select subset used by Utils::pseudo_inverse
struct BoolArr;
template <class T> struct Arr {
    Dyn<T> v;
    Arr<T> abs() const { Arr<T> o{v}; for (auto &e : o.v.d) e = std::abs(e); return o; }
    Arr<T> inverse() const { Arr<T> o{v}; for (auto &e : o.v.d) e = T(1) / e; return o; }
    T operator()(Index i) const { return v.d[(size_t)i]; }
    Dyn<T> matrix() const { return v; }
    BoolArr operator>(T t) const;
};
struct BoolArr {
    std::vector<char> b; Index r, c;
    template <class T, class S> Arr<T> select(const Arr<T> &then_, S else_) const { Arr<T> o{then_.v}; for (size_t k = 0; k < b.size(); ++k) if (!b[k]) o.v.d[k] = (T)else_; return o; }
};
template <class T> BoolArr Arr<T>::operator>(T t) const { BoolArr o; o.r = v.r; o.c = v.c; o.b.resize(v.d.size()); for (size_t k = 0; k < v.d.size(); ++k) o.b[k] = v.d[k] > t; return o; }
template <class T> Arr<T> Dyn<T>::array() const { return Arr<T>{*this}; }

// one-sided Jacobi SVD (small square matrices): A = U S V', singular values sorted in decreasing order
template <class M> class JacobiSVD {
    DynD U, V; VectorXd S;
  public:
    JacobiSVD(const DynD &A, unsigned = 0) {
        Index n = A.r; shim_check(A.r == A.c, "JacobiSVD: square only");
        DynD W = A; V.resize(n, n); V.setIdentity();
        for (int sweep = 0; sweep < 60; ++sweep) {
            double off = 0;
            for (Index p = 0; p < n; ++p) for (Index q = p + 1; q < n; ++q) {
                double a = 0, b = 0, g = 0;
                for (Index i = 0; i < n; ++i) { a += W(i, p) * W(i, p); b += W(i, q) * W(i, q); g += W(i, p) * W(i, q); }
                if (std::abs(g) <= 1e-300 || std::abs(g) <= std::numeric_limits<double>::epsilon() * std::sqrt(a * b)) continue;
                off = std::max(off, std::abs(g) / std::sqrt(a * b));
                double zeta = (b - a) / (2 * g), t = (zeta >= 0 ? 1.0 : -1.0) / (std::abs(zeta) + std::sqrt(1 + zeta * zeta));
                double cs = 1 / std::sqrt(1 + t * t), sn = cs * t;
                for (Index i = 0; i < n; ++i) { double wp = W(i, p), wq = W(i, q); W(i, p) = cs * wp - sn * wq; W(i, q) = sn * wp + cs * wq; double vp = V(i, p), vq = V(i, q); V(i, p) = cs * vp - sn * vq; V(i, q) = sn * vp + cs * vq; }
            }
            if (off == 0) break;
        }
        S.resize(n); U.resize(n, n);
        std::vector<Index> ord((size_t)n);
        for (Index j = 0; j < n; ++j) { double s = 0; for (Index i = 0; i < n; ++i) s += W(i, j) * W(i, j); S(j) = std::sqrt(s); ord[(size_t)j] = j; }
        for (Index a = 0; a < n; ++a) for (Index b = a + 1; b < n; ++b) if (S(ord[(size_t)b]) > S(ord[(size_t)a])) std::swap(ord[(size_t)a], ord[(size_t)b]);
        DynD W2 = W, V2 = V; VectorXd S2 = S;
        for (Index j = 0; j < n; ++j) { Index o = ord[(size_t)j]; S(j) = S2(o); for (Index i = 0; i < n; ++i) { W(i, j) = W2(i, o); V(i, j) = V2(i, o); } }
        // U: normalised columns of W; columns of (numerically) zero singular values completed to an orthonormal basis
        for (Index j = 0; j < n; ++j) {
            if (S(j) > 1e-300) { for (Index i = 0; i < n; ++i) U(i, j) = W(i, j) / S(j); }
            else { for (Index i = 0; i < n; ++i) U(i, j) = 0; }
        }
        for (Index j = 0; j < n; ++j) {
            double nn = 0; for (Index i = 0; i < n; ++i) nn += U(i, j) * U(i, j);
            if (nn > 0.5) continue;
            for (Index e = 0; e < n; ++e) {       // Gram-Schmidt a unit vector against the columns already set
                std::vector<double> v((size_t)n, 0.0); v[(size_t)e] = 1;
                for (Index k = 0; k < n; ++k) { if (k == j) continue; double nk = 0, dot = 0; for (Index i = 0; i < n; ++i) { nk += U(i, k) * U(i, k); dot += U(i, k) * v[(size_t)i]; } if (nk > 0.5) for (Index i = 0; i < n; ++i) v[(size_t)i] -= dot * U(i, k); }
                double vn = 0; for (double x : v) vn += x * x;
                if (vn > 1e-6) { vn = std::sqrt(vn); for (Index i = 0; i < n; ++i) U(i, j) = v[(size_t)i] / vn; break; }
            }
        }
    }
    const VectorXd &singularValues() const { return S; }
    const DynD &matrixU() const { return U; }
    const DynD &matrixV() const { return V; }
};

// ---------------------------------------------------------------------------------------------------------
// geometry subset
template <class T> class AngleAxis;
template <class T> class Quaternion {
    Matrix<T, 4, 1> q;   // x y z w (Eigen's coefficient order)
  public:
    Quaternion() { q.d[3] = 1; }
    Quaternion(T w, T x, T y, T z) { q.d[0] = x; q.d[1] = y; q.d[2] = z; q.d[3] = w; }
    Quaternion(const AngleAxis<T> &aa);
    void setIdentity() { q.d[0] = q.d[1] = q.d[2] = 0; q.d[3] = 1; }
    Matrix<T, 4, 1> &coeffs() { return q; }
    const Matrix<T, 4, 1> &coeffs() const { return q; }
    T x() const { return q.d[0]; } T y() const { return q.d[1]; } T z() const { return q.d[2]; } T w() const { return q.d[3]; }
    Matrix<T, 3, 3> toRotationMatrix() const {
        Matrix<T, 3, 3> R; T tx = 2 * x(), ty = 2 * y(), tz = 2 * z(), twx = tx * w(), twy = ty * w(), twz = tz * w(), txx = tx * x(), txy = ty * x(), txz = tz * x(), tyy = ty * y(), tyz = tz * y(), tzz = tz * z();
        R(0, 0) = 1 - (tyy + tzz); R(0, 1) = txy - twz; R(0, 2) = txz + twy; R(1, 0) = txy + twz; R(1, 1) = 1 - (txx + tzz); R(1, 2) = tyz - twx; R(2, 0) = txz - twy; R(2, 1) = tyz + twx; R(2, 2) = 1 - (txx + tyy);
        return R;
    }
    Matrix<T, 3, 3> matrix() const { return toRotationMatrix(); }
    Quaternion operator*(const Quaternion &b) const {
        const Quaternion &a = *this;
        return Quaternion(a.w() * b.w() - a.x() * b.x() - a.y() * b.y() - a.z() * b.z(), a.w() * b.x() + a.x() * b.w() + a.y() * b.z() - a.z() * b.y(),
                          a.w() * b.y() + a.y() * b.w() + a.z() * b.x() - a.x() * b.z(), a.w() * b.z() + a.z() * b.w() + a.x() * b.y() - a.y() * b.x());
    }
    Quaternion operator*(const AngleAxis<T> &b) const { return *this * Quaternion(b); }
    static Quaternion UnitRandom() { T u1 = std::rand() / (T)RAND_MAX, u2 = 2 * M_PI * (std::rand() / (T)RAND_MAX), u3 = 2 * M_PI * (std::rand() / (T)RAND_MAX); T a = std::sqrt(1 - u1), b = std::sqrt(u1); return Quaternion(a * std::sin(u2), a * std::cos(u2), b * std::sin(u3), b * std::cos(u3)); }
};
template <class T> class AngleAxis {
  public:
    T ang; Matrix<T, 3, 1> ax;
    AngleAxis(T a, const Matrix<T, 3, 1> &axis) : ang(a), ax(axis) {}
    Quaternion<T> operator*(const AngleAxis &b) const { return Quaternion<T>(*this) * Quaternion<T>(b); }
};
template <class T> Quaternion<T>::Quaternion(const AngleAxis<T> &aa) { T h = aa.ang / 2, s = std::sin(h); q.d[0] = s * aa.ax.d[0]; q.d[1] = s * aa.ax.d[1]; q.d[2] = s * aa.ax.d[2]; q.d[3] = std::cos(h); }
typedef Quaternion<double> Quaterniond;
typedef AngleAxis<double> AngleAxisd;

template <class T> Dyn<T> Dyn<T>::eulerAngles(int, int, int) const { throw std::runtime_error("mini_eigen: eulerAngles not provided"); }

}  // namespace Eigen
