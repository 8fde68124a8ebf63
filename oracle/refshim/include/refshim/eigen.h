/*
 * TEST INFRASTRUCTURE -- stand-in for the few Eigen 3.3.7 types the unmodified reference sources use (Eigen::Matrix4d, a row-major
 * dynamic Map converted into it, operator(), inverse()).  Eigen is NOT in /root/reference (un-vendored, pinned only by the Docker base):
 * Matrix4d::inverse() is restated here from knowledge of Eigen/src/LU/arch/Inverse_SSE.h (double specialisation, what an x86-64 Release
 * build without -march runs for Session.cpp:110 / RosParamServer.cpp:30) -- PARITY UNPINNED, the same operation order as the oracle's
 * orc_inverse4x4 and the product's ltm_inverse4x4.  Original code.
 */
#ifndef REFSHIM_EIGEN_H
#define REFSHIM_EIGEN_H

#include <cstddef>
#include <memory>

#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#define EIGEN_ALIGN16 alignas(16)

namespace Eigen {

enum { ColMajor = 0, RowMajor = 1, Dynamic = -1 };

template <class T> using aligned_allocator = std::allocator<T>;

template <class S, int R, int C, int Opt = ColMajor> class Matrix;
template <class M> class Map;

/* Map<const Matrix<double,-1,-1,RowMajor>>(ptr, rows, cols): Session.cpp:108, RosParamServer.cpp:29 */
template <> class Map<const Matrix<double, Dynamic, Dynamic, RowMajor>> {
public:
    Map(const double* p, int rows, int cols) : p_(p), rows_(rows), cols_(cols) {}
    double operator()(int r, int c) const { return p_[(size_t)r * cols_ + c]; }
    int rows() const { return rows_; }
    int cols() const { return cols_; }
private:
    const double* p_;
    int rows_, cols_;
};

template <> class Matrix<double, 4, 4, ColMajor> {
public:
    Matrix() {}                                     /* Eigen leaves a fixed-size matrix uninitialised */
    Matrix(const Map<const Matrix<double, Dynamic, Dynamic, RowMajor>>& src)
    {
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) (*this)(r, c) = src(r, c);
    }
    double& operator()(int r, int c) { return m_[c * 4 + r]; }                 /* column-major storage, like Eigen's default */
    const double& operator()(int r, int c) const { return m_[c * 4 + r]; }
    const double* data() const { return m_; }
    static Matrix Identity()
    {
        Matrix I;
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) I(r, c) = r == c ? 1.0 : 0.0;
        return I;
    }

    /* general 4x4 inverse, 2x2-block form: the storage read in memory order as blocks A B / C D of N = M^T; X# = adjugate;
     * AB = A#B, DC = D#C, det = |A||D| + |B||C| - trace(AB DC); blocks (A|D| - B DC)#, (C|B| - D AB#)#, (B|C| - A DC#)#,
     * (D|A| - C AB)# times +-1/det; every product and every sum rounded on its own (SSE2, no FMA). */
    Matrix inverse() const
    {
        const double* s = m_;                     /* column-major M read in memory order: s[4*r + k] = M(k, r) = N(r, k), N = M^T */
        double A[2][2], B[2][2], C[2][2], D[2][2];
        for (int r = 0; r < 2; ++r)
            for (int k = 0; k < 2; ++k) {
                A[r][k] = s[4 * r + k];     B[r][k] = s[4 * r + k + 2];
                C[r][k] = s[4 * (r + 2) + k]; D[r][k] = s[4 * (r + 2) + k + 2];
            }
        const double dA = A[0][0] * A[1][1] - A[0][1] * A[1][0], dB = B[0][0] * B[1][1] - B[0][1] * B[1][0];
        const double dC = C[0][0] * C[1][1] - C[0][1] * C[1][0], dD = D[0][0] * D[1][1] - D[0][1] * D[1][0];
        double AB[2][2], DC[2][2], iA[2][2], iB[2][2], iC[2][2], iD[2][2];
        for (int j = 0; j < 2; ++j) {
            AB[0][j] = B[0][j] * A[1][1] - B[1][j] * A[0][1]; AB[1][j] = B[1][j] * A[0][0] - B[0][j] * A[1][0];
            DC[0][j] = C[0][j] * D[1][1] - C[1][j] * D[0][1]; DC[1][j] = C[1][j] * D[0][0] - C[0][j] * D[1][0];
        }
        const double tr = (AB[0][0] * DC[0][0] + AB[1][0] * DC[0][1]) + (AB[0][1] * DC[1][0] + AB[1][1] * DC[1][1]);
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j) {
                const double cab = AB[0][j] * C[i][0] + AB[1][j] * C[i][1];
                const double bdc = DC[0][j] * B[i][0] + DC[1][j] * B[i][1];
                iD[i][j] = D[i][j] * dA - cab;
                iA[i][j] = A[i][j] * dD - bdc;
            }
        for (int i = 0; i < 2; ++i) {
            iB[i][0] = D[i][0] * AB[1][1] - D[i][1] * AB[1][0]; iB[i][1] = D[i][1] * AB[0][0] - D[i][0] * AB[0][1];
            iC[i][0] = A[i][0] * DC[1][1] - A[i][1] * DC[1][0]; iC[i][1] = A[i][1] * DC[0][0] - A[i][0] * DC[0][1];
        }
        const double d1 = dA * dD, d2 = dB * dC;
        const double det = (d1 + d2) - tr;
        const double rd = 1.0 / det, nrd = -rd;
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j) { iB[i][j] = C[i][j] * dB - iB[i][j]; iC[i][j] = B[i][j] * dC - iC[i][j]; }
        Matrix out;
        double* o = out.m_;                        /* o[4*r + k] = Ninv(r, k) = Minv(k, r): the same memory convention */
        const double (*blk[4])[2] = {iA, iB, iC, iD};
        const int at[4][2] = {{0, 0}, {0, 2}, {2, 0}, {2, 2}};
        for (int q = 0; q < 4; ++q) {
            const double (*X)[2] = blk[q];
            const int r = at[q][0], c = at[q][1];
            o[4 * r + c] = X[1][1] * rd;       o[4 * r + c + 1] = X[0][1] * nrd;
            o[4 * (r + 1) + c] = X[1][0] * nrd; o[4 * (r + 1) + c + 1] = X[0][0] * rd;
        }
        return out;
    }

private:
    double m_[16];
};

typedef Matrix<double, 4, 4, ColMajor> Matrix4d;

} // namespace Eigen

#endif
