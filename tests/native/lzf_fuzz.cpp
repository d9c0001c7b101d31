// AddressSanitizer / UBSan driver for aln_lzf_decompress (autolabel_amd/csrc/capi.cpp): random and damaged streams against
// exactly-sized heap buffers, so that any read past the input or write past the output aborts the run.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
extern "C" int64_t aln_lzf_decompress(const void* src, size_t n_in, void* dst, size_t n_out);
extern "C" const char* aln_last_error(void);

static uint64_t s = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 11); }

// a valid LZF stream for `n` bytes of compressible data (literal runs and back references, including overlapping ones)
static size_t make_stream(unsigned char* out, unsigned char* plain, size_t n) {
  size_t op = 0, ip = 0;
  while (ip < n) {
    if (ip >= 8 && (rnd() & 1)) {
      size_t dist = 1 + rnd() % (ip < 8192 ? ip : 8192), len = 3 + rnd() % 262;
      if (len > n - ip) len = n - ip;
      if (len < 3) goto literal;
      for (size_t k = 0; k < len; ++k) plain[ip + k] = plain[ip + k - dist];
      size_t l = len - 2;
      if (l < 7) out[op++] = (unsigned char)((l << 5) | ((dist - 1) >> 8));
      else { out[op++] = (unsigned char)((7 << 5) | ((dist - 1) >> 8)); out[op++] = (unsigned char)(l - 7); }
      out[op++] = (unsigned char)((dist - 1) & 255);
      ip += len;
      continue;
    }
  literal: {
      size_t run = 1 + rnd() % 32;
      if (run > n - ip) run = n - ip;
      out[op++] = (unsigned char)(run - 1);
      for (size_t k = 0; k < run; ++k) { plain[ip] = (unsigned char)rnd(); out[op++] = plain[ip++]; }
    }
  }
  return op;
}

int main() {
  int ok = 0, rejected = 0;
  for (int trial = 0; trial < 20000; ++trial) {
    size_t n = 1 + rnd() % 3000;
    unsigned char* plain = (unsigned char*)malloc(n);
    unsigned char* tmp = (unsigned char*)malloc(2 * n + 16);
    size_t m = make_stream(tmp, plain, n);
    unsigned char* in = (unsigned char*)malloc(m);      // exact size: ASan sees the first byte read past the stream
    memcpy(in, tmp, m);
    int mode = trial % 4;
    size_t n_in = m, n_out = n;
    if (mode == 1) n_in = rnd() % (m + 1);                                  // truncated stream
    if (mode == 2) n_out = rnd() % (n + 1);                                 // output buffer too small
    if (mode == 3) for (int k = 0; k < 3; ++k) in[rnd() % m] = (unsigned char)rnd();   // damaged stream
    unsigned char* out = (unsigned char*)malloc(n_out ? n_out : 1);
    int64_t got = aln_lzf_decompress(in, n_in, out, n_out);
    if (mode == 0) {
      if (got != (int64_t)n || memcmp(out, plain, n)) { printf("MISMATCH trial %d\n", trial); return 1; }
      ++ok;
    } else if (got < 0) ++rejected;
    free(plain); free(tmp); free(in); free(out);
  }
  printf("lzf fuzz: %d round trips, %d damaged streams rejected\n", ok, rejected);
  return ok == 5000 && rejected > 1000 ? 0 : 1;
}
