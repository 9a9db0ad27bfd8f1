// output_tools.h -- PAF records and writer (reference: PAFMapping output_tools.h:16-38,
// AppendUnmappedRead :200-210, PAFOutputTools<PAFMapping>::AppendMapping :336-354).
#ifndef SIGMAP_AMD_OUTPUTTOOLS_H_
#define SIGMAP_AMD_OUTPUTTOOLS_H_
#include <cstdint>
#include <cstdio>
#include <string>

namespace sigmap {
struct PAFMapping {
  uint32_t read_id;
  std::string read_name;
  uint32_t read_length;
  uint32_t read_start_position;
  uint32_t read_end_position;
  uint32_t fragment_start_position;
  uint32_t fragment_length;
  uint8_t mapq;       // 0..60 mapped, 61 = unmapped row
  uint8_t direction;  // 1 = +
  uint8_t is_unique;
  std::string tags;
};

// 12 PAF columns + tags; rows with mapq > 60 print "*" in columns 3-11.
inline std::string FormatPAF(const PAFMapping &m, const char *ref_name, uint32_t ref_length) {
  if (m.mapq <= 60) {
    return m.read_name + "\t" + std::to_string(m.read_length) + "\t" +
           std::to_string(m.read_start_position) + "\t" + std::to_string(m.read_end_position) + "\t" +
           ((m.direction & 1) ? "+" : "-") + "\t" + ref_name + "\t" + std::to_string(ref_length) + "\t" +
           std::to_string(m.fragment_start_position) + "\t" +
           std::to_string(m.fragment_start_position + m.fragment_length) + "\t" +
           std::to_string(m.read_length) + "\t" + std::to_string(m.fragment_length) + "\t" +
           std::to_string(m.mapq) + "\t" + m.tags + "\n";
  }
  return m.read_name + "\t" + std::to_string(m.read_length) + "\t*\t*\t*\t*\t*\t*\t*\t*\t*\t" +
         std::to_string(m.mapq) + "\t" + m.tags + "\n";
}
}  // namespace sigmap
#endif
