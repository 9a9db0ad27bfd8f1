// utils.h -- small host helpers (mirrors the few pieces of the reference's utils.h that the
// hot path's callers use: utils.h:53-71 timers/exit, :73-96 base tables, :98-116 k-mer hash).
#ifndef SIGMAP_AMD_UTILS_H_
#define SIGMAP_AMD_UTILS_H_
#include <sys/time.h>

#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <string>

namespace sigmap {
inline double GetRealTime() {
  struct timeval tp;
  gettimeofday(&tp, nullptr);
  return tp.tv_sec + tp.tv_usec * 1e-6;
}
inline void ExitWithMessage(const std::string &message) {
  std::cerr << message << std::endl;
  exit(-1);
}
// A,C,G,T (either case) -> 0..3, everything else 4
inline uint8_t CharToUint8(const char c) {
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
  }
}
inline char Uint8ToChar(const uint8_t i) { return i < 4 ? "ACGT"[i] : 'N'; }
// 2-bit k-mer hash of sequence[start, start+seed_length); ambiguous bases and positions at
// or beyond sequence_length count as A.
inline uint64_t GenerateSeedFromSequence(const char *sequence, size_t sequence_length,
                                         uint32_t start_position, uint32_t seed_length) {
  const uint64_t mask = (((uint64_t)1) << (2 * seed_length)) - 1;
  uint64_t seed = 0;
  for (uint32_t i = 0; i < seed_length; ++i) {
    uint8_t b = 4;
    if ((size_t)start_position + i < sequence_length) b = CharToUint8(sequence[start_position + i]);
    seed = ((seed << 2) | (b < 4 ? b : 0)) & mask;
  }
  return seed;
}
}  // namespace sigmap
#endif
