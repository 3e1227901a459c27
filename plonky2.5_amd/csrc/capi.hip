// C ABI of libp25 (declared in include/p25.h).  Thin: argument checks, device buffers, launches.
#include <cstring>
#include <memory>
#include <mutex>
#include "../../include/p25.h"
#include "kernels.h"

namespace p25 {
thread_local std::string g_last_error;
static bool g_device_ok = false;
static std::unique_ptr<NttTables> g_tables;

NttTables& tables() {
  if (!g_tables) g_tables.reset(new NttTables());
  return *g_tables;
}

p25_status ensure_device() {
  if (g_device_ok) return P25_OK;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    g_last_error = "no HIP device available (libp25 has no CPU fallback)";
    return P25_ERR_NO_DEVICE;
  }
  g_device_ok = true;
  return P25_OK;
}

struct DevBuf {
  u64* p = nullptr;
  explicit DevBuf(size_t words) {
    if (words) P25_HIP(hipMalloc(&p, words * sizeof(u64)));
  }
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
};

template <class F>
p25_status guarded(F&& f) {
  try {
    p25_status s = ensure_device();
    if (s != P25_OK) return s;
    return f();
  } catch (const HipError& e) {
    g_last_error = e.what();
    return P25_ERR_HIP;
  } catch (const std::invalid_argument& e) {
    g_last_error = e.what();
    return P25_ERR_INVALID_ARG;
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return P25_ERR_INTERNAL;
  }
}

static bool is_pow2(size_t x) { return x && !(x & (x - 1)); }

void lde_commit_dev(const u64* d_polys, unsigned log_n, size_t n_polys, bool from_coeffs,
                    unsigned rate_bits, unsigned cap_height, u64* d_coeffs, u64* d_tmp, u64* d_lde,
                    u64* d_tree, hipStream_t st) {
  const size_t n = (size_t)1 << log_n;
  const u64* coeffs = d_polys;
  if (!from_coeffs) {
    ntt_inverse(tables(), d_polys, n, false, d_tmp, n, d_coeffs, n, (int)log_n, (int)n_polys, 1, st);
    coeffs = d_coeffs;
  }
  const size_t big = n << rate_bits;
  ntt_lde_bitrev(tables(), coeffs, n, d_lde, big, (int)log_n, (int)rate_bits, (int)n_polys,
                 gl::GENERATOR, st);
  if (d_tree) launch_merkle_tree(d_lde, big, (int)n_polys, big, cap_height, d_tree, st);
}
}  // namespace p25

using namespace p25;

extern "C" {

const char* p25_last_error(void) { return g_last_error.c_str(); }
const char* p25_version(void) { return "libp25 0.1 (gfx950)"; }

p25_status p25_device_init(int device_index) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    g_last_error = "no HIP device available (libp25 has no CPU fallback)";
    return P25_ERR_NO_DEVICE;
  }
  if (device_index < 0 || device_index >= n) {
    g_last_error = "device index out of range";
    return P25_ERR_INVALID_ARG;
  }
  if (hipSetDevice(device_index) != hipSuccess) {
    g_last_error = "hipSetDevice failed";
    return P25_ERR_HIP;
  }
  g_device_ok = true;
  return P25_OK;
}

p25_status p25_poseidon_permute(uint64_t* states, size_t n) {
  return guarded([&]() -> p25_status {
    if (!states && n) throw std::invalid_argument("states is null");
    DevBuf d(n * 12);
    P25_HIP(hipMemcpy(d.p, states, n * 96, hipMemcpyHostToDevice));
    launch_poseidon_permute(d.p, n, 0);
    P25_HIP(hipGetLastError());
    P25_HIP(hipMemcpy(states, d.p, n * 96, hipMemcpyDeviceToHost));
    return P25_OK;
  });
}

p25_status p25_poseidon2_permute(uint64_t* states, size_t n) {
  return guarded([&]() -> p25_status {
    if (!states && n) throw std::invalid_argument("states is null");
    DevBuf d(n * 12);
    P25_HIP(hipMemcpy(d.p, states, n * 96, hipMemcpyHostToDevice));
    launch_poseidon2_permute(d.p, n, 0);
    P25_HIP(hipGetLastError());
    P25_HIP(hipMemcpy(states, d.p, n * 96, hipMemcpyDeviceToHost));
    return P25_OK;
  });
}

size_t p25_merkle_tree_words(size_t n_leaves, unsigned cap_height) {
  if (!is_pow2(n_leaves) || cap_height > 63 || n_leaves < ((size_t)1 << cap_height)) return 0;
  return merkle_tree_words(n_leaves, cap_height);
}

p25_status p25_merkle_commit(const uint64_t* leaves_cm, size_t n_leaves, size_t width,
                             unsigned cap_height, uint64_t* cap_out, uint64_t* tree_out) {
  return guarded([&]() -> p25_status {
    if (!is_pow2(n_leaves) || cap_height > 40 || n_leaves < ((size_t)1 << cap_height) || !width ||
        width > (1u << 20) || !leaves_cm)
      throw std::invalid_argument("p25_merkle_commit: bad shape");
    DevBuf d(n_leaves * width);
    size_t tw = merkle_tree_words(n_leaves, cap_height);
    DevBuf t(tw);
    P25_HIP(hipMemcpy(d.p, leaves_cm, n_leaves * width * 8, hipMemcpyHostToDevice));
    u64* cap = launch_merkle_tree(d.p, n_leaves, (int)width, n_leaves, cap_height, t.p, 0);
    P25_HIP(hipGetLastError());
    if (cap_out) P25_HIP(hipMemcpy(cap_out, cap, ((size_t)32) << cap_height, hipMemcpyDeviceToHost));
    if (tree_out) P25_HIP(hipMemcpy(tree_out, t.p, tw * 8, hipMemcpyDeviceToHost));
    P25_HIP(hipDeviceSynchronize());
    return P25_OK;
  });
}

p25_status p25_merkle_commit_dev(const uint64_t* d_leaves_cm, size_t col_stride, size_t n_leaves,
                                 size_t width, unsigned cap_height, uint64_t* d_tree, void* stream) {
  return guarded([&]() -> p25_status {
    if (!is_pow2(n_leaves) || cap_height > 40 || n_leaves < ((size_t)1 << cap_height) || !width ||
        col_stride < n_leaves || !d_leaves_cm || !d_tree)
      throw std::invalid_argument("p25_merkle_commit_dev: bad shape");
    launch_merkle_tree(d_leaves_cm, col_stride, (int)width, n_leaves, cap_height, d_tree,
                       (hipStream_t)stream);
    P25_HIP(hipGetLastError());
    return P25_OK;
  });
}

p25_status p25_poseidon_permute_dev(uint64_t* d_states, size_t n, void* stream) {
  return guarded([&]() -> p25_status {
    launch_poseidon_permute(d_states, n, (hipStream_t)stream);
    P25_HIP(hipGetLastError());
    return P25_OK;
  });
}

p25_status p25_lde_commit(const uint64_t* polys, unsigned log_n, size_t n_polys, int from_coeffs,
                          unsigned rate_bits, unsigned cap_height, uint64_t* coeffs_out,
                          uint64_t* lde_out, uint64_t* cap_out) {
  return guarded([&]() -> p25_status {
    if (!polys || !n_polys || log_n > 20 || rate_bits > 3 || log_n + rate_bits < cap_height)
      throw std::invalid_argument("p25_lde_commit: bad shape");
    const size_t n = (size_t)1 << log_n, big = n << rate_bits;
    DevBuf in(n * n_polys), co(n * n_polys), tmp(n * n_polys), lde(big * n_polys);
    size_t tw = merkle_tree_words(big, cap_height);
    DevBuf tree(tw);
    P25_HIP(hipMemcpy(in.p, polys, n * n_polys * 8, hipMemcpyHostToDevice));
    lde_commit_dev(in.p, log_n, n_polys, from_coeffs != 0, rate_bits, cap_height, co.p, tmp.p, lde.p,
                   tree.p, 0);
    P25_HIP(hipGetLastError());
    P25_HIP(hipDeviceSynchronize());
    if (coeffs_out)
      P25_HIP(hipMemcpy(coeffs_out, from_coeffs ? in.p : co.p, n * n_polys * 8, hipMemcpyDeviceToHost));
    if (lde_out) P25_HIP(hipMemcpy(lde_out, lde.p, big * n_polys * 8, hipMemcpyDeviceToHost));
    if (cap_out)
      P25_HIP(hipMemcpy(cap_out, tree.p + tw - ((size_t)4 << cap_height), ((size_t)32) << cap_height,
                        hipMemcpyDeviceToHost));
    return P25_OK;
  });
}

p25_status p25_lde_commit_dev(const uint64_t* d_polys, unsigned log_n, size_t n_polys, int from_coeffs,
                              unsigned rate_bits, unsigned cap_height, uint64_t* d_coeffs,
                              uint64_t* d_tmp, uint64_t* d_lde, uint64_t* d_tree, void* stream) {
  return guarded([&]() -> p25_status {
    if (!d_polys || !n_polys || log_n > 20 || rate_bits > 3 || !d_lde ||
        (!from_coeffs && (!d_coeffs || !d_tmp)))
      throw std::invalid_argument("p25_lde_commit_dev: bad shape");
    lde_commit_dev(d_polys, log_n, n_polys, from_coeffs != 0, rate_bits, cap_height, d_coeffs, d_tmp,
                   d_lde, d_tree, (hipStream_t)stream);
    P25_HIP(hipGetLastError());
    return P25_OK;
  });
}

}  // extern "C"
