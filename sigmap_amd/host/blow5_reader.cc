// blow5_reader.cc -- BLOW5 ingestion for SignalBatch.
//
// The reference reads BLOW5 through slow5lib (signal_batch.cc:156-210: slow5_open, slow5_get_next
// until SLOW5_ERR_EOF, and of every record only read_id, digitisation, offset, range,
// len_raw_signal, raw_signal).  slow5lib (hasindu2008/slow5lib, an un-vendored submodule: no pinned
// revision in the checkout) is not in this image, so the binary layout of the published SLOW5
// specification is parsed directly:
//
//   file   : "BLOW5\1" | u8 major, minor, patch | u8 record compression (0 none, 1 zlib, 2 zstd) [byte 9]
//            | u32 number of read groups [bytes 10-13] | [version >= 0.2.0: u8 signal compression (0 none,
//            1 svb-zd), byte 14 -- taken from what was padding before] | zero padding up to byte 64
//            | u32 header_size | ASCII SLOW5 header (header_size bytes; its
//            last line names the columns: #read_id read_group digitisation offset range sampling_rate
//            len_raw_signal raw_signal [auxiliary fields...]) | records | "5WOLB"
//   record : u64 record_size | record_size bytes, compressed as a whole when record compression is
//            set; plain: u16 read_id_len | read_id | u32 read_group | f64 digitisation | f64 offset |
//            f64 range | f64 sampling_rate | u64 len_raw_signal | raw_signal | auxiliary fields (ignored
//            here, as in the reference); raw_signal = i16[len_raw_signal], or with svb-zd signal
//            compression: u32 n (= len_raw_signal) | ceil(n / 4) control bytes | data bytes -- classic
//            StreamVByte over 32-bit words (a 2-bit code per value, first value in the low bits of a control
//            byte: 1 + code little-endian data bytes), the words being the zig-zag codes
//            ((d << 1) ^ (d >> 31)) of the differences d between consecutive samples widened to 32 bits, the
//            first one against 0
//   (little endian throughout)
//
// Supported: record compression none / zlib (/ zstd when the library is present at build time), signal
// compression none / svb-zd (slow5tools' default is zlib + svb-zd).  digitisation / offset / range are narrowed to float exactly where the
// reference does it (`digitisation = rec->digitisation`, signal_batch.cc:187-189); the pA loop is the
// same AddSignal as for FAST5 (signal_batch.cc:195-207).  A file without the end-of-file marker is an
// error, like slow5_get_next returning anything but SLOW5_ERR_EOF (signal_batch.cc:170-173).
//
// WITHOUT slow5lib THIS READER IS CHECKED ONLY AGAINST THE WRITER BELOW AND THE TESTS' INDEPENDENT PYTHON WRITER,
// both written from the same recollection of slow5lib's layout (slow5_hdr_to_mem / slow5_rec_to_mem /
// ptr_compress_svb_zd): no file produced by slow5lib or slow5tools has ever been read by it.
#include <zlib.h>

#include <cstdio>
#include <cstring>

#include "signal_batch.h"
#include "utils.h"

#ifdef SMHOST_HAVE_ZSTD
#include <zstd.h>
#endif

namespace sigmap {
namespace {
const char kMagic[6] = {'B', 'L', 'O', 'W', '5', '\1'};
const char kEof[5] = {'5', 'W', 'O', 'L', 'B'};
const char *kColumns = "#read_id\tread_group\tdigitisation\toffset\trange\tsampling_rate\tlen_raw_signal\traw_signal";

template <typename T>
bool Take(const std::vector<unsigned char> &buf, size_t *pos, T *out) {
  if (*pos + sizeof(T) > buf.size()) return false;
  memcpy(out, buf.data() + *pos, sizeof(T));
  *pos += sizeof(T);
  return true;
}

bool InflateRecord(const std::vector<unsigned char> &in, std::vector<unsigned char> *out) {
  z_stream zs;
  memset(&zs, 0, sizeof zs);
  if (inflateInit(&zs) != Z_OK) return false;
  out->resize(std::max<size_t>(in.size() * 4, 1024));
  zs.next_in = const_cast<unsigned char *>(in.data());
  zs.avail_in = (uInt)in.size();
  size_t produced = 0;
  int rc = Z_OK;
  while (rc != Z_STREAM_END) {
    if (produced == out->size()) out->resize(out->size() * 2);
    zs.next_out = out->data() + produced;
    zs.avail_out = (uInt)std::min<size_t>(out->size() - produced, 1u << 30);
    rc = inflate(&zs, Z_NO_FLUSH);
    produced = zs.total_out;
    if (rc != Z_OK && rc != Z_STREAM_END && rc != Z_BUF_ERROR) break;
    if (rc == Z_BUF_ERROR && zs.avail_in == 0 && produced < out->size()) break;  // truncated stream
  }
  inflateEnd(&zs);
  out->resize(produced);
  return rc == Z_STREAM_END;
}

// svb-zd (see the file header): n values from at most `avail` bytes; false when the stream is short or its count
// differs from the record's len_raw_signal.
bool DecodeSvbZd(const unsigned char *p, size_t avail, size_t n_expected, std::vector<int16_t> *out) {
  uint32_t n = 0;
  if (avail < 4) return false;
  memcpy(&n, p, 4);
  if (n != n_expected) return false;
  const size_t n_ctrl = ((size_t)n + 3) / 4;
  if (avail - 4 < n_ctrl) return false;
  const unsigned char *ctrl = p + 4, *data = ctrl + n_ctrl, *end = p + avail;
  out->resize(n);
  int32_t prev = 0;
  for (size_t i = 0; i < n; ++i) {
    const unsigned bytes = ((ctrl[i >> 2] >> (2 * (i & 3))) & 3u) + 1u;
    if ((size_t)(end - data) < bytes) return false;
    uint32_t v = 0;
    for (unsigned b = 0; b < bytes; ++b) v |= (uint32_t)data[b] << (8 * b);
    data += bytes;
    const int32_t d = (int32_t)(v >> 1) ^ -(int32_t)(v & 1u);  // zig-zag
    prev += d;
    (*out)[i] = (int16_t)prev;
  }
  return true;
}

void EncodeSvbZd(const std::vector<int16_t> &raw, std::vector<unsigned char> *out) {
  const uint32_t n = (uint32_t)raw.size();
  const size_t n_ctrl = ((size_t)n + 3) / 4, base = out->size();
  out->resize(base + 4 + n_ctrl, 0);
  memcpy(out->data() + base, &n, 4);
  int32_t prev = 0;
  for (size_t i = 0; i < n; ++i) {
    const int32_t d = (int32_t)raw[i] - prev;
    prev = raw[i];
    const uint32_t v = ((uint32_t)d << 1) ^ (uint32_t)(d >> 31);
    const unsigned bytes = v < (1u << 8) ? 1u : v < (1u << 16) ? 2u : v < (1u << 24) ? 3u : 4u;
    (*out)[base + 4 + (i >> 2)] |= (unsigned char)((bytes - 1u) << (2 * (i & 3)));
    for (unsigned b = 0; b < bytes; ++b) out->push_back((unsigned char)(v >> (8 * b)));
  }
}

template <typename T>
void Put(std::vector<unsigned char> *buf, const T &v) {
  const unsigned char *p = reinterpret_cast<const unsigned char *>(&v);
  buf->insert(buf->end(), p, p + sizeof(T));
}
}  // namespace

bool SignalBatch::HaveZstd() {
#ifdef SMHOST_HAVE_ZSTD
  return true;
#else
  return false;
#endif
}

void SignalBatch::AddSignalsFromBLOW5(const std::string &path) {
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) ExitWithMessage("Error in opening file " + path);  // signal_batch.cc:158-162
  unsigned char head[64];
  if (fread(head, 1, 64, f) != 64 || memcmp(head, kMagic, 6) != 0) ExitWithMessage("Not a BLOW5 file: " + path);
  const unsigned major = head[6], minor = head[7];
  const unsigned record_press = head[9];
  // bytes 10-13: number of read groups (not needed: the records carry their group, which the reference ignores)
  const bool has_signal_press = major > 0 || minor >= 2;  // the byte was added in file version 0.2.0, at offset 14
  const unsigned signal_press = has_signal_press ? head[14] : 0;
  if (record_press > 2) ExitWithMessage("BLOW5 " + path + ": unknown record compression " + std::to_string(record_press));
  if (record_press == 2 && !HaveZstd())
    ExitWithMessage("BLOW5 " + path + ": zstd-compressed records, but this build has no zstd (Makefile: ZSTD_ROOT)");
  if (signal_press > 1)
    ExitWithMessage("BLOW5 " + path + ": signal compression " + std::to_string(signal_press) +
                    " is not supported by this reader (none and svb-zd are); rewrite the file with `slow5tools view -s svb-zd`");
  uint32_t header_size = 0;
  if (fread(&header_size, 4, 1, f) != 1) ExitWithMessage("Truncated BLOW5 header: " + path);
  std::string header(header_size, '\0');
  if (header_size && fread(&header[0], 1, header_size, f) != header_size) ExitWithMessage("Truncated BLOW5 header: " + path);
  // the column line: the primary fields in the order the records hold them
  const size_t col = header.rfind("#read_id");
  if (col == std::string::npos || header.compare(col, strlen(kColumns), kColumns) != 0)
    ExitWithMessage("BLOW5 " + path + ": unexpected column line (primary SLOW5 fields expected)");
  std::vector<unsigned char> packed, rec;
  std::vector<int16_t> raw;
  for (;;) {
    unsigned char sz[8];
    const size_t got = fread(sz, 1, 8, f);
    if (got == 5 && memcmp(sz, kEof, 5) == 0) break;  // proper end of file
    if (got != 8) {  // signal_batch.cc:170-173: anything but a clean EOF is fatal
      ExitWithMessage("Error in reading BLOW5 records (no end-of-file marker): " + path);
    }
    if (memcmp(sz, kEof, 5) == 0) {  // the marker followed by stray bytes
      ExitWithMessage("Error in reading BLOW5 records (data after the end-of-file marker): " + path);
    }
    uint64_t record_size = 0;
    memcpy(&record_size, sz, 8);
    if (record_size == 0 || record_size > ((uint64_t)1 << 34)) ExitWithMessage("Corrupt BLOW5 record size in " + path +
                      " (this reader parses the binary SLOW5 layout itself and has never been checked against a slow5lib-written file: if slow5tools reads the file, report it)");
    packed.resize((size_t)record_size);
    if (fread(packed.data(), 1, packed.size(), f) != packed.size()) ExitWithMessage("Truncated BLOW5 record in " + path);
    const std::vector<unsigned char> *body = &packed;
    if (record_press == 1) {
      if (!InflateRecord(packed, &rec)) ExitWithMessage("BLOW5 " + path + ": a zlib record does not inflate");
      body = &rec;
    } else if (record_press == 2) {
#ifdef SMHOST_HAVE_ZSTD
      const unsigned long long want = ZSTD_getFrameContentSize(packed.data(), packed.size());
      if (want == ZSTD_CONTENTSIZE_ERROR || want == ZSTD_CONTENTSIZE_UNKNOWN || want > ((uint64_t)1 << 34))
        ExitWithMessage("BLOW5 " + path + ": a zstd record without a usable content size");
      rec.resize((size_t)want);
      const size_t n = ZSTD_decompress(rec.data(), rec.size(), packed.data(), packed.size());
      if (ZSTD_isError(n) || n != rec.size()) ExitWithMessage("BLOW5 " + path + ": a zstd record does not decompress");
      body = &rec;
#endif
    }
    size_t pos = 0;
    uint16_t id_len = 0;
    uint32_t read_group = 0;
    double digitisation = 0, offset = 0, range = 0, sampling_rate = 0;
    uint64_t len_raw = 0;
    bool ok = Take(*body, &pos, &id_len) && pos + id_len <= body->size();
    std::string id;
    if (ok) {
      id.assign(reinterpret_cast<const char *>(body->data() + pos), id_len);
      pos += id_len;
      // (a writer may count the terminating NUL in read_id_len)
      while (!id.empty() && id.back() == '\0') id.pop_back();
    }
    ok = ok && Take(*body, &pos, &read_group) && Take(*body, &pos, &digitisation) && Take(*body, &pos, &offset) &&
         Take(*body, &pos, &range) && Take(*body, &pos, &sampling_rate) && Take(*body, &pos, &len_raw);
    if (ok && signal_press == 1) {
      ok = len_raw <= ((uint64_t)1 << 32) && DecodeSvbZd(body->data() + pos, body->size() - pos, (size_t)len_raw, &raw);
    } else {
      ok = ok && len_raw <= (body->size() - pos) / sizeof(int16_t);
      if (ok) {
        raw.resize((size_t)len_raw);
        if (len_raw) memcpy(raw.data(), body->data() + pos, (size_t)len_raw * sizeof(int16_t));
      }
    }
    if (!ok)
      ExitWithMessage("Corrupt BLOW5 record in " + path +
                      " (this reader parses the binary SLOW5 layout itself and has never been checked against a slow5lib-written file: if slow5tools reads the file, report it)");
    // signal_batch.cc:186-189: float digitisation = rec->digitisation (double -> float), likewise range, offset
    AddSignal(id, (float)digitisation, (float)range, (float)offset, raw.data(), raw.size());
  }
  fclose(f);
}

bool SignalBatch::WriteBLOW5(const std::string &path, int record_press, int signal_press, const std::vector<std::string> &ids,
                             double digitisation, double range, double offset,
                             const std::vector<std::vector<int16_t>> &raws) {
  if (record_press < 0 || record_press > 2 || (record_press == 2 && !HaveZstd()) || signal_press < 0 || signal_press > 1) return false;
  FILE *f = fopen(path.c_str(), "wb");
  if (!f) return false;
  unsigned char head[64];
  memset(head, 0, sizeof head);
  memcpy(head, kMagic, 6);
  head[6] = 0;  // file version 0.2.0
  head[7] = 2;
  head[8] = 0;
  head[9] = (unsigned char)record_press;
  const uint32_t n_groups = 1;
  memcpy(head + 10, &n_groups, 4);
  head[14] = (unsigned char)signal_press;  // 0 none, 1 svb-zd
  std::string header = "#slow5_version\t0.2.0\n#num_read_groups\t1\n@asic_id\tsynthetic\n"
                       "#char*\tuint32_t\tdouble\tdouble\tdouble\tdouble\tuint64_t\tint16_t*\n";
  header += kColumns;
  header += "\n";
  const uint32_t header_size = (uint32_t)header.size();
  bool ok = fwrite(head, 1, 64, f) == 64 && fwrite(&header_size, 4, 1, f) == 1 &&
            fwrite(header.data(), 1, header.size(), f) == header.size();
  std::vector<unsigned char> rec, packed;
  for (size_t r = 0; ok && r < ids.size(); ++r) {
    rec.clear();
    Put(&rec, (uint16_t)ids[r].size());
    rec.insert(rec.end(), ids[r].begin(), ids[r].end());
    Put(&rec, (uint32_t)0);
    Put(&rec, digitisation);
    Put(&rec, offset);
    Put(&rec, range);
    Put(&rec, (double)4000.0);
    Put(&rec, (uint64_t)raws[r].size());
    if (signal_press == 1) {
      EncodeSvbZd(raws[r], &rec);
    } else {
      const unsigned char *s = reinterpret_cast<const unsigned char *>(raws[r].data());
      rec.insert(rec.end(), s, s + raws[r].size() * sizeof(int16_t));
    }
    const std::vector<unsigned char> *body = &rec;
    if (record_press == 1) {
      uLongf cap = compressBound((uLong)rec.size());
      packed.resize(cap);
      if (compress2(packed.data(), &cap, rec.data(), (uLong)rec.size(), Z_DEFAULT_COMPRESSION) != Z_OK) ok = false;
      packed.resize(cap);
      body = &packed;
    } else if (record_press == 2) {
#ifdef SMHOST_HAVE_ZSTD
      packed.resize(ZSTD_compressBound(rec.size()));
      const size_t n = ZSTD_compress(packed.data(), packed.size(), rec.data(), rec.size(), 1);
      if (ZSTD_isError(n)) ok = false;
      else packed.resize(n);
      body = &packed;
#endif
    }
    const uint64_t record_size = body->size();
    ok = ok && fwrite(&record_size, 8, 1, f) == 1 && fwrite(body->data(), 1, body->size(), f) == body->size();
  }
  ok = ok && fwrite(kEof, 1, 5, f) == 5;
  return (fclose(f) == 0) && ok;
}
}  // namespace sigmap
