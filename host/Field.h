// Mesh-node containers with the reference's access surface (cpp/Field.h:55-218 of
// jcoady/MultiGrid): a 3-D field indexed f[i][j][k] through T***, k contiguous,
// zero-filled at construction, public const ni/nj/nk.  Written from scratch for the
// HIP host adapter; only what the multigrid path and its driver touch is provided.
//
// Unlike the reference (one heap block per (i,j) row) the node data is by default ONE block
// in [i][j][k] order with two pointer tables on top, so the whole field can go to the device
// library without a packing pass.  Code that only uses f[i][j][k], f(i,j,k), clear() and
// ni/nj/nk cannot tell the difference.  Compiled with -DMG_FIELD_ROW_PER_ALLOCATION the
// rows are separate heap blocks as in the reference (cpp/Field.h:65-70): the layout a
// solver meets when it is built into the reference program.
#ifndef MG_HOST_FIELD_H
#define MG_HOST_FIELD_H

#include <cstddef>
#include <new>
#include <ostream>
#include <vector>

template <typename T>
struct vec3 {
    vec3() : d{T(), T(), T()} {}
    vec3(T a, T b, T c) : d{a, b, c} {}
    T &operator[](int a) { return d[a]; }
    T operator()(int a) const { return d[a]; }
    vec3 &operator=(double s) { d[0] = d[1] = d[2] = T(s); return *this; }

protected:
    T d[3];
};
template <typename T> vec3<T> operator+(const vec3<T> &a, const vec3<T> &b) { return {a(0) + b(0), a(1) + b(1), a(2) + b(2)}; }
template <typename T> vec3<T> operator-(const vec3<T> &a, const vec3<T> &b) { return {a(0) - b(0), a(1) - b(1), a(2) - b(2)}; }
template <typename T> vec3<T> operator*(T s, const vec3<T> &a) { return {s * a(0), s * a(1), s * a(2)}; }
template <typename T> std::ostream &operator<<(std::ostream &o, const vec3<T> &v) { return o << v(0) << " " << v(1) << " " << v(2); }

using double3 = vec3<double>;
using int3 = vec3<int>;

// Where the node blocks of contiguous Fields come from.  Null (the default) means operator new;
// a driver may point the pair at a page-locked allocator (mg_host_alloc / mg_host_free of
// libmg_hip) BEFORE it constructs its World, so that host <-> device copies of the fields run at
// the full host-link rate.  The pair must stay set until the last Field is gone.
inline void *(*field_block_alloc)(std::size_t bytes) = nullptr;
inline void (*field_block_free)(void *) = nullptr;

template <typename T>
class Field_ {
public:
    Field_(int ni_, int nj_, int nk_) : ni(ni_), nj(nj_), nk(nk_), rows(std::size_t(ni_) * nj_), planes(ni_)
    {
#ifdef MG_FIELD_ROW_PER_ALLOCATION
        // the reference's allocation pattern (cpp/Field.h:65-70): one zero-filled heap block per
        // (i,j) row, so rows are NOT adjacent in memory
        store.resize(rows.size());
        for (std::size_t r = 0; r < rows.size(); r++) {
            store[r].assign(std::size_t(nk) + r % 3, T());  // uneven block sizes keep rows apart
            rows[r] = store[r].data();
        }
#else
        const std::size_t count = std::size_t(ni_) * nj_ * nk_;
        if (field_block_alloc) {
            cells = static_cast<T *>(field_block_alloc(count * sizeof(T)));
            release = field_block_free;
        }
        if (!cells) {
            cells = static_cast<T *>(::operator new(count * sizeof(T)));
            release = nullptr;
        }
        for (std::size_t p = 0; p < count; p++) new (cells + p) T();
        for (std::size_t r = 0; r < rows.size(); r++) rows[r] = cells + r * nk;
#endif
        for (int i = 0; i < ni; i++) planes[i] = rows.data() + std::size_t(i) * nj;
        data = planes.data();
    }
    Field_(const Field_ &o) : Field_(o.ni, o.nj, o.nk)
    {
        for (std::size_t r = 0; r < rows.size(); r++)
            for (int k = 0; k < nk; k++) rows[r][k] = o.rows[r][k];
    }
    Field_ &operator=(const Field_ &) = delete;
#ifndef MG_FIELD_ROW_PER_ALLOCATION
    ~Field_()
    {
        if (release) release(cells);
        else ::operator delete(cells);
    }
#endif

    T **operator[](int i) { return data[i]; }                          // f[i][j][k]
    T operator()(int i, int j, int k) const { return data[i][j][k]; }  // read-only access
    void operator=(double s)
    {
        for (T *r : rows)
            for (int k = 0; k < nk; k++) r[k] = T(s);
    }
    void clear() { (*this) = 0; }
    Field_ &operator+=(const Field_ &o)
    {
        for (std::size_t r = 0; r < rows.size(); r++)
            for (int k = 0; k < nk; k++) rows[r][k] += o.rows[r][k];
        return *this;
    }
    Field_ &operator*=(double s)
    {
        for (T *r : rows)
            for (int k = 0; k < nk; k++) r[k] *= s;
        return *this;
    }
    // logical node number with i fastest (cpp/Field.h:161); not the memory order
    int U(int i, int j, int k) const { return k * ni * nj + j * ni + i; }

    std::size_t size() const { return std::size_t(ni) * nj * nk; }
#ifndef MG_FIELD_ROW_PER_ALLOCATION
    // the node data in memory order ((i*nj)+j)*nk+k (not part of the reference's surface)
    T *flat() { return cells; }
    const T *flat() const { return cells; }
#endif

    const int ni, nj, nk;

protected:
#ifdef MG_FIELD_ROW_PER_ALLOCATION
    std::vector<std::vector<T>> store;
#else
    T *cells = nullptr;  // trivially destructible node types only (double, int, vec3)
    void (*release)(void *) = nullptr;
#endif
    std::vector<T *> rows;
    std::vector<T **> planes;
    T ***data;
};

using Field = Field_<double>;
using FieldI = Field_<int>;
using Field3 = Field_<double3>;

#endif
