/* Host-side helpers of the NETCDF4 reader (dmd_era5_amd/hdf5_lite.py); plain C, no GPU code.
 * HDF5 hands variable-length strings back as one malloc'ed C string per element; label
 * coordinates such as `original_variable (space)` (ref era5_svd.py:300-333) have 10^6..10^7
 * of them, and a Python loop over them costs seconds.  These two passes turn the pointer
 * array into a fixed-width byte matrix that numpy can take over. */
#include <stddef.h>
#include <string.h>

size_t dmdx_host_vlen_maxlen(const char* const* p, size_t n) {
  size_t mx = 0;
  for (size_t i = 0; i < n; ++i) {
    if (p[i]) {
      size_t l = strlen(p[i]);
      if (l > mx) mx = l;
    }
  }
  return mx;
}

/* out: n rows of `width` bytes, zero padded; returns 1 if every byte is 7-bit ASCII */
int dmdx_host_vlen_to_fixed(const char* const* p, size_t n, char* out, size_t width) {
  unsigned char acc = 0;
  memset(out, 0, n * width);
  for (size_t i = 0; i < n; ++i) {
    if (!p[i]) continue;
    size_t l = strlen(p[i]);
    if (l > width) l = width;
    char* o = out + i * width;
    for (size_t j = 0; j < l; ++j) {
      o[j] = p[i][j];
      acc |= (unsigned char)p[i][j];
    }
  }
  return acc < 128;
}
