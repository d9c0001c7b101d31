// Error reporting and ABI version for libautolabel_hip.
#include <stdarg.h>
#include <stdio.h>
#include "../../include/autolabel_hip.h"

static thread_local char g_err[512] = "";

void aln_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* aln_last_error(void) { return g_err; }
extern "C" int aln_abi_version(void) { return ALN_ABI_VERSION; }

// LZF decoder (liblzf's published stream format: control byte < 32 = literal run, else a back reference of 3..264
// bytes at distance 1..8192) for the chunks of features.hdf (h5py filter 32000).  Host code: no device work.
extern "C" int64_t aln_lzf_decompress(const void* src_, size_t n_in, void* dst_, size_t n_out) {
  const unsigned char* ip = (const unsigned char*)src_;
  const unsigned char* const in_end = ip + n_in;
  unsigned char* op = (unsigned char*)dst_;
  unsigned char* const out0 = op;
  unsigned char* const out_end = op + n_out;
  while (ip < in_end) {
    unsigned ctrl = *ip++;
    if (ctrl < 32) {
      const size_t run = ctrl + 1;
      if ((size_t)(out_end - op) < run || (size_t)(in_end - ip) < run) { aln_set_error("lzf: literal run overflows"); return -1; }
      for (size_t k = 0; k < run; ++k) op[k] = ip[k];
      op += run, ip += run;
    } else {
      size_t len = ctrl >> 5;
      size_t dist = ((size_t)(ctrl & 31) << 8) + 1;
      if (len == 7) {
        if (ip >= in_end) { aln_set_error("lzf: truncated stream"); return -1; }
        len += *ip++;
      }
      if (ip >= in_end) { aln_set_error("lzf: truncated stream"); return -1; }
      dist += *ip++;
      len += 2;
      if ((size_t)(op - out0) < dist || (size_t)(out_end - op) < len) { aln_set_error("lzf: bad back reference"); return -1; }
      const unsigned char* ref = op - dist;
      for (size_t k = 0; k < len; ++k) op[k] = ref[k];   // byte order matters: overlapping references replicate
      op += len;
    }
  }
  return (int64_t)(op - out0);
}
