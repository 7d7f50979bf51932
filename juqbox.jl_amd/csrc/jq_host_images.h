// jq_host_images.h -- part of the host side of libjuqbox_hip.so (included by juqbox_hip.hip, ONE translation unit; not a stand-alone header):
// operator / state images in the kernels' layouts and the structure tests the planner uses.
// A-fragment tile image of a column-major Ntot x Ntot matrix: only the tiles of the block band
// |mt - kk/4| <= BW are stored, in walk order (kk outer, mt inner); tile (mt,kk) lane l holds
// M[16*mt + (l&15)][4*kk + (l>>4)]; zero padded.
static void tile_image(const double* M, int Ntot, int NT, int BW, double* img, bool SD = false)
{
    if (BW == JQ_BW_T4) {
        // compact image (JQ_T4_ELEMS doubles): per 16-row block the 4x4 diagonal blocks of its four 4-row groups rho = 4 mt + b,
        // element JQ_T4_AIDX(rho, k, i) = M[4 rho + i][4 rho + k] (= lane 16 k + 4 b + i of the quad-layout MFMA's A operand), then the
        // coupling coefficients
        // per block mt: JQ_T4_CIDX(g, r, term: group rho-1, rho+1 (same 16-row block), rho-4, rho+4) <-> row 4 rho + g, rho = 4 mt + r
        const int NR = 4 * NT;
        for (size_t i = 0; i < (size_t)JQ_T4_ELEMS(NT); ++i) img[i] = 0.0;
        if (!SD)
            for (int rho = 0; rho < NR; ++rho)
                for (int k = 0; k < 4; ++k)
                    for (int i = 0; i < 4; ++i) {
                        const int row = 4 * rho + i, col = 4 * rho + k;
                        img[JQ_T4_AIDX(rho, k, i)] = (row < Ntot && col < Ntot) ? M[row + (size_t)Ntot * col] : 0.0;
                    }
        double* cf = img + (size_t)NR * JQ_T4_TILE;
        for (int rho = 0; rho < NR; ++rho)
            for (int g = 0; g < 4; ++g) {
                const int row = 4 * rho + g, r = rho & 3;
                const int nbr[4] = {r > 0 ? row - 4 : -1, r < 3 ? row + 4 : -1, row - 16, row + 16};
                for (int t = 0; t < 4; ++t) {
                    const int col = nbr[t];
                    cf[(rho >> 2) * 64 + JQ_T4_CIDX(g, r, t)] = (col >= 0 && col < Ntot && row < Ntot) ? M[row + (size_t)Ntot * col] : 0.0;
                }
            }
        return;
    }
    const int KT = 4 * NT;
    size_t idx = 0;
    for (int kk = 0; kk < KT; ++kk)
        for (int mt = 0; mt < NT; ++mt) {
            const int kb = kk >> 2;
            if (!block_on(BW, SD, mt, kb)) continue;
            for (int l = 0; l < 64; ++l) {
                const int row = 16 * mt + (l & 15), col = 4 * kk + (l >> 4);
                img[idx * 64 + l] = (row < Ntot && col < Ntot) ? M[row + (size_t)Ntot * col] : 0.0;
            }
            ++idx;
        }
    if (BW == JQ_BW_OD) {
        // diagonals of the first off-diagonal blocks: [mt][dir: block mt-1, block mt+1][g][r] <-> row 16mt + 4r + g
        double* cf = img + idx * 64;
        for (int mt = 0; mt < NT; ++mt)
            for (int dir = 0; dir < 2; ++dir) {
                const int nb = mt + (dir ? 1 : -1);
                for (int g = 0; g < 4; ++g)
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * mt + 4 * r + g, col = 16 * nb + 4 * r + g;
                        cf[((mt * 2 + dir) * 4 + g) * 4 + r] =
                            (nb >= 0 && nb < NT && row < Ntot && col < Ntot) ? M[row + (size_t)Ntot * col] : 0.0;
                    }
            }
    }
}

// JQ_BW_T4 structure: entries outside the 4x4 diagonal blocks only at (i, i +- 4) inside one 16-row block or at (i, i +- 16)
static bool t4_structure(const double* M, int Ntot)
{
    for (int col = 0; col < Ntot; ++col)
        for (int row = 0; row < Ntot; ++row) {
            if (M[row + (size_t)Ntot * col] == 0.0 || row / 4 == col / 4) continue;
            const int d = row - col;
            const bool same16 = (row / 16 == col / 16);
            if (!((same16 && (d == 4 || d == -4)) || d == 16 || d == -16)) return false;
        }
    return true;
}
// parts of the T4 image of M that are non-zero: JQ_T4_DIAG | JQ_T4_RTERMS | JQ_T4_MTERMS
static int t4_mode(const double* M, int Ntot)
{
    int mode = 0;
    for (int col = 0; col < Ntot; ++col)
        for (int row = 0; row < Ntot; ++row) {
            if (M[row + (size_t)Ntot * col] == 0.0) continue;
            if (row / 4 == col / 4) mode |= JQ_T4_DIAG;
            else if (row - col == 4 || col - row == 4) mode |= JQ_T4_RTERMS;
            else mode |= JQ_T4_MTERMS;
        }
    return mode;
}

// true if M is block tridiagonal (16x16 blocks) and every off-diagonal block is a diagonal matrix
static bool offdiag_blocks_diagonal(const double* M, int Ntot)
{
    for (int col = 0; col < Ntot; ++col)
        for (int row = 0; row < Ntot; ++row) {
            if (M[row + (size_t)Ntot * col] == 0.0) continue;
            const int d = row / 16 - col / 16;
            if (d == 0) continue;
            if (std::abs(d) > 1 || row % 16 != col % 16) return false;
        }
    return true;
}

// Row-window layout of the cooperative kernels (jq_coop_kernels.h): for tile row mt the NB k-blocks
// kb0(mt)..kb0(mt)+NB-1, 4 tiles each, rows consecutively.
static void tile_image_coop(const double* M, int Ntot, int NT, int BW, double* img)
{
    const int NB = coop_nb(NT, BW);
    if (BW == JQ_BW_OD) {
        // per tile row: the 4 tiles of the diagonal block, then [dir: block mt-1, mt+1][g][r] <-> row 16mt + 4r + g
        for (int mt = 0; mt < NT; ++mt) {
            double* row = img + (size_t)mt * coop_row_elems(NT, BW);
            for (int r4 = 0; r4 < 4; ++r4)
                for (int l = 0; l < 64; ++l) {
                    const int rr = 16 * mt + (l & 15), col = 4 * (4 * mt + r4) + (l >> 4);
                    row[r4 * 64 + l] = (rr < Ntot && col < Ntot) ? M[rr + (size_t)Ntot * col] : 0.0;
                }
            for (int dir = 0; dir < 2; ++dir) {
                const int nb = mt + (dir ? 1 : -1);
                for (int g = 0; g < 4; ++g)
                    for (int r = 0; r < 4; ++r) {
                        const int rr = 16 * mt + 4 * r + g, col = 16 * nb + 4 * r + g;
                        row[256 + (dir * 4 + g) * 4 + r] = (nb >= 0 && nb < NT && rr < Ntot && col < Ntot) ? M[rr + (size_t)Ntot * col] : 0.0;
                    }
            }
        }
        return;
    }
    size_t idx = 0;
    for (int mt = 0; mt < NT; ++mt) {
        const int kb0 = coop_kb0(NT, BW, mt);
        for (int j = 0; j < NB; ++j)
            for (int r = 0; r < 4; ++r) {
                const int kk = 4 * (kb0 + j) + r;
                for (int l = 0; l < 64; ++l) {
                    const int row = 16 * mt + (l & 15), col = 4 * kk + (l >> 4);
                    img[idx * 64 + l] = (row < Ntot && col < Ntot) ? M[row + (size_t)Ntot * col] : 0.0;
                }
                ++idx;
            }
    }
}

// true if every diagonal 16x16 block of M is zero
static bool diag_blocks_zero(const double* M, int Ntot)
{
    for (int col = 0; col < Ntot; ++col)
        for (int row = 0; row < Ntot; ++row)
            if (row / 16 == col / 16 && M[row + (size_t)Ntot * col] != 0.0) return false;
    return true;
}

// smallest block band width that contains every nonzero of M
static int block_band(const double* M, int Ntot)
{
    int bw = 0;
    for (int col = 0; col < Ntot; ++col)
        for (int row = 0; row < Ntot; ++row)
            if (M[row + (size_t)Ntot * col] != 0.0) bw = std::max(bw, std::abs(row / 16 - col / 16));
    return bw;
}

// register-layout images [parts][KT][64] of an Ntot x N array: N <= 16: one image, the N columns replicated over the sps
// samples of a slab; N > 16: part p holds the columns 16 p .. 16 p + 15
static void slab_image(const double* A, int Ntot, int N, int sps, int KT, double* img, int parts = 1)
{
    for (int p = 0; p < parts; ++p)
        for (int kk = 0; kk < KT; ++kk)
            for (int l = 0; l < 64; ++l) {
                const int row = 4 * kk + (l >> 4), col = l & 15;
                const int scol = parts > 1 ? 16 * p + col : col % N;
                const bool on = parts > 1 ? scol < N : col < sps * N;
                img[((size_t)p * KT + kk) * 64 + l] = (row < Ntot && on) ? A[row + (size_t)Ntot * scol] : 0.0;
            }
}

// plain row-major NP x NP image of a column-major Ntot x Ntot matrix (lane kernels), zero padded
static void plain_image(const double* M, int Ntot, int NP, double* img)
{
    for (int i = 0; i < Ntot; ++i)
        for (int j = 0; j < Ntot; ++j) img[(size_t)i * NP + j] = M[i + (size_t)Ntot * j];
}

// [N][NP] image of an Ntot x N array (lane kernels)
static void column_image(const double* A, int Ntot, int N, int NP, double* img)
{
    for (int c = 0; c < N; ++c)
        for (int r = 0; r < Ntot; ++r) img[(size_t)c * NP + r] = A[r + (size_t)Ntot * c];
}

// [16][NPJ] row-major image of a column-major Ntot x Ntot matrix (row-lane kernels), zero padded
// Dense policy of the cooperative-quad kernels (jq_cq_kernels.h CoopQ<2, true>; 17 <= Ntot <= 32): per 16-row block mt eight operand
// registers of 64 lanes -- rotation s = 0 .. 3 of the block's own tile (mt, mt), then of the tile (mt, 1 - mt).  Lane 16 k + 4 b + i of
// rotation s holds M[16 mt + 4 b + i][16 t' + 4 ((b - s) mod 4) + k]: the A operand of the v_mfma_f64_4x4x4_4b whose B operand is the state
// register of block t' rotated by 4 s lanes inside each 16-lane row.  Rows / columns beyond Ntot: zero.
#define JQ_DQ_ELEMS (2 * 8 * 64)
static void dq_image(const double* M, int Ntot, double* img)
{
    std::fill_n(img, JQ_DQ_ELEMS, 0.0);
    for (int mt = 0; mt < 2; ++mt)
        for (int half = 0; half < 2; ++half) {
            const int tp = half == 0 ? mt : 1 - mt;
            for (int s = 0; s < 4; ++s)
                for (int k = 0; k < 4; ++k)
                    for (int b = 0; b < 4; ++b)
                        for (int i = 0; i < 4; ++i) {
                            const int row = 16 * mt + 4 * b + i, col = 16 * tp + 4 * ((b - s + 4) % 4) + k;
                            if (row < Ntot && col < Ntot) img[(size_t)((mt * 8 + half * 4 + s) * 64) + 16 * k + 4 * b + i] = M[row + (size_t)Ntot * col];
                        }
        }
}
static void rowlane_image(const double* M, int Ntot, int NPJ, double* img)
{
    for (int i = 0; i < Ntot; ++i)
        for (int j = 0; j < Ntot; ++j) img[(size_t)i * NPJ + j] = M[i + (size_t)Ntot * j];
}

