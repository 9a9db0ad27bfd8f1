// fast5_reader.cc -- FAST5 ingestion for SignalBatch.
//
// What the reference reads from a FAST5 file (signal_batch.cc:69-151, through its vendored
// hdf5_tools.hpp), restated on the plain HDF5 C API:
//  * format: a root group named "Raw" means single-read layout -- every group under /Raw/Reads is
//    a read, channel attributes in /UniqueGlobalKey/channel_id; otherwise EVERY root group is a
//    read (multi-read layout): /<read>/Raw and /<read>/channel_id (signal_batch.cc:73-104);
//  * groups are listed by H5Lget_name_by_idx(H5_INDEX_NAME, H5_ITER_NATIVE) (hdf5_tools.hpp:2071-2093);
//  * attributes are fetched AS STRINGS and parsed with atof (signal_batch.cc:110-133): a float
//    attribute is read as double and printed with operator<< at the stream's default precision
//    (hdf5_tools.hpp:1124-1141), so digitisation / range / offset keep SIX significant digits --
//    reproduced here, because it changes the pA values the whole pipeline sees;
//  * Raw/Signal is read as float, then pA = (raw + offset) * (range / digitisation), kept iff
//    30 < pA < 200 (signal_batch.cc:135-149).
// Files are taken in sorted path order (the reference follows readdir order, which is not defined).
//
// Built only when the HDF5 headers and library are present (Makefile: SMHOST_HAVE_HDF5); otherwise
// a FAST5 file is a loud error.  The writer exists for the tests and the synthetic generator.
#include <cstdlib>
#include <cstring>
#include <sstream>

#include "signal_batch.h"
#include "utils.h"

#ifdef SMHOST_HAVE_HDF5
#include <hdf5.h>

namespace sigmap {
namespace {
std::vector<std::string> ListGroup(hid_t file, const std::string &path) {
  std::vector<std::string> names;
  hid_t g = H5Gopen2(file, path.c_str(), H5P_DEFAULT);
  if (g < 0) return names;
  H5G_info_t info;
  if (H5Gget_info(g, &info) >= 0) {
    for (hsize_t i = 0; i < info.nlinks; ++i) {
      const ssize_t sz = H5Lget_name_by_idx(file, path.c_str(), H5_INDEX_NAME, H5_ITER_NATIVE, i, nullptr, 0, H5P_DEFAULT);
      if (sz < 0) break;
      std::string name((size_t)sz, '\0');
      H5Lget_name_by_idx(file, path.c_str(), H5_INDEX_NAME, H5_ITER_NATIVE, i, &name[0], (size_t)sz + 1, H5P_DEFAULT);
      names.push_back(name);
    }
  }
  H5Gclose(g);
  return names;
}

// An attribute's value as the string the reference's reader would hand to atof / use as the id.
bool AttrAsString(hid_t file, const std::string &obj, const char *name, std::string *out) {
  if (H5Aexists_by_name(file, obj.c_str(), name, H5P_DEFAULT) <= 0) return false;
  hid_t a = H5Aopen_by_name(file, obj.c_str(), name, H5P_DEFAULT, H5P_DEFAULT);
  if (a < 0) return false;
  hid_t t = H5Aget_type(a);
  const H5T_class_t cls = H5Tget_class(t);
  bool ok = true;
  std::ostringstream oss;
  if (cls == H5T_STRING) {
    if (H5Tis_variable_str(t) > 0) {
      char *p = nullptr;
      hid_t mt = H5Tcopy(H5T_C_S1);
      H5Tset_size(mt, H5T_VARIABLE);
      H5Tset_cset(mt, H5Tget_cset(t));
      ok = H5Aread(a, mt, &p) >= 0 && p != nullptr;
      if (ok) oss << p;
      if (p) free(p);  // allocated by libhdf5 with the system allocator
      H5Tclose(mt);
    } else {
      const size_t sz = H5Tget_size(t);
      std::vector<char> buf(sz + 1, '\0');
      hid_t mt = H5Tcopy(H5T_C_S1);
      H5Tset_size(mt, sz + 1);
      H5Tset_cset(mt, H5Tget_cset(t));
      ok = H5Aread(a, mt, buf.data()) >= 0;
      std::string v(buf.data(), sz);
      while (!v.empty() && v.back() == '\0') v.pop_back();
      oss << v;
      H5Tclose(mt);
    }
  } else if (cls == H5T_INTEGER) {
    if (H5Tget_sign(t) == H5T_SGN_NONE) {
      unsigned long long v = 0;
      ok = H5Aread(a, H5T_NATIVE_ULLONG, &v) >= 0;
      oss << v;
    } else {
      long long v = 0;
      ok = H5Aread(a, H5T_NATIVE_LLONG, &v) >= 0;
      oss << v;
    }
  } else if (cls == H5T_FLOAT) {
    double v = 0;
    ok = H5Aread(a, H5T_NATIVE_DOUBLE, &v) >= 0;
    oss << v;  // default precision: six significant digits, like the reference
  } else {
    ok = false;
  }
  H5Tclose(t);
  H5Aclose(a);
  if (ok) *out = oss.str();
  return ok;
}

void WriteStringAttr(hid_t obj, const char *name, const std::string &v) {
  hid_t t = H5Tcopy(H5T_C_S1);
  H5Tset_size(t, v.size() > 0 ? v.size() : 1);
  hid_t sp = H5Screate(H5S_SCALAR);
  hid_t a = H5Acreate2(obj, name, t, sp, H5P_DEFAULT, H5P_DEFAULT);
  H5Awrite(a, t, v.c_str());
  H5Aclose(a);
  H5Sclose(sp);
  H5Tclose(t);
}
void WriteDoubleAttr(hid_t obj, const char *name, double v) {
  hid_t sp = H5Screate(H5S_SCALAR);
  hid_t a = H5Acreate2(obj, name, H5T_IEEE_F64LE, sp, H5P_DEFAULT, H5P_DEFAULT);
  H5Awrite(a, H5T_NATIVE_DOUBLE, &v);
  H5Aclose(a);
  H5Sclose(sp);
}
void WriteChannel(hid_t g, float digitisation, float range, float offset) {
  WriteStringAttr(g, "channel_number", "1");
  WriteDoubleAttr(g, "digitisation", digitisation);
  WriteDoubleAttr(g, "offset", offset);
  WriteDoubleAttr(g, "range", range);
  WriteDoubleAttr(g, "sampling_rate", 4000.0);
}
void WriteSignal(hid_t g, const std::vector<int16_t> &raw) {
  hsize_t dims[1] = {raw.size()};
  hid_t sp = H5Screate_simple(1, dims, nullptr);
  hid_t d = H5Dcreate2(g, "Signal", H5T_STD_I16LE, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
  H5Dwrite(d, H5T_NATIVE_SHORT, H5S_ALL, H5S_ALL, H5P_DEFAULT, raw.data());
  H5Dclose(d);
  H5Sclose(sp);
}
}  // namespace

bool SignalBatch::HaveFAST5() { return true; }

void SignalBatch::AddSignalsFromFAST5(const std::string &path) {
  H5Eset_auto2(H5E_DEFAULT, nullptr, nullptr);  // errors are reported here, not by libhdf5's stack dump
  hid_t file = H5Fopen(path.c_str(), H5F_ACC_RDONLY, H5P_DEFAULT);
  if (file < 0) ExitWithMessage("Cannot open FAST5 file " + path);
  const std::vector<std::string> root = ListGroup(file, "/");
  bool is_single = false;
  for (const std::string &g : root) is_single = is_single || g == "Raw";
  auto add_read = [&](const std::string &raw_path, const std::string &ch_path) {
    std::string id, v;
    AttrAsString(file, raw_path, "read_id", &id);
    float digitisation = 0, range = 0, offset = 0;
    if (AttrAsString(file, ch_path, "digitisation", &v)) digitisation = (float)atof(v.c_str());
    if (AttrAsString(file, ch_path, "range", &v)) range = (float)atof(v.c_str());
    if (AttrAsString(file, ch_path, "offset", &v)) offset = (float)atof(v.c_str());
    const std::string sig_path = raw_path + "/Signal";
    hid_t d = H5Dopen2(file, sig_path.c_str(), H5P_DEFAULT);
    if (d < 0) ExitWithMessage("FAST5 read without " + sig_path + " in " + path);
    hid_t sp = H5Dget_space(d);
    const hssize_t npts = H5Sget_simple_extent_npoints(sp);
    if (raw_mode_) {
      // the ADC counts stay 16-bit integers and the device converts them (SignalBatch::KeepRaw); MinKNOW
      // writes the Signal dataset as int16, anything wider cannot be kept exactly
      hid_t ft = H5Dget_type(d);
      const bool int16_ok = H5Tget_class(ft) == H5T_INTEGER &&
                            (H5Tget_size(ft) < 2 || (H5Tget_size(ft) == 2 && H5Tget_sign(ft) == H5T_SGN_2));
      H5Tclose(ft);
      if (!int16_ok) ExitWithMessage(sig_path + " in " + path + " is not a 16-bit signed integer dataset");
      std::vector<int16_t> raw(npts > 0 ? (size_t)npts : 0);
      if (npts > 0 && H5Dread(d, H5T_NATIVE_SHORT, H5S_ALL, H5S_ALL, H5P_DEFAULT, raw.data()) < 0)
        ExitWithMessage("Cannot read " + sig_path + " in " + path);
      H5Sclose(sp);
      H5Dclose(d);
      AddSignal(id, digitisation, range, offset, raw.data(), raw.size());
      return;
    }
    std::vector<float> values(npts > 0 ? (size_t)npts : 0);
    if (npts > 0 && H5Dread(d, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, values.data()) < 0)
      ExitWithMessage("Cannot read " + sig_path + " in " + path);
    H5Sclose(sp);
    H5Dclose(d);
    Signal s;
    s.id = id;
    s.digitisation = digitisation;
    s.range = range;
    s.offset = offset;
    s.signal_values.reserve(values.size());
    const float scale = range / digitisation;
    for (float x : values) {
      const float pa = (x + offset) * scale;
      if (pa > 30 && pa < 200) s.signal_values.push_back(pa);
    }
    signals_.push_back(std::move(s));
  };
  if (is_single) {
    for (const std::string &read : ListGroup(file, "/Raw/Reads")) {
      std::string id;
      if (!AttrAsString(file, "/Raw/Reads/" + read, "read_id", &id) || id.empty())
        ExitWithMessage("Error: failed to find read_id in " + path);
      add_read("/Raw/Reads/" + read, "/UniqueGlobalKey/channel_id");
    }
  } else {
    for (const std::string &read : root) add_read("/" + read + "/Raw", "/" + read + "/channel_id");
  }
  H5Fclose(file);
}

bool SignalBatch::WriteFAST5(const std::string &path, bool multi_read, const std::vector<std::string> &ids,
                             float digitisation, float range, float offset,
                             const std::vector<std::vector<int16_t>> &raws) {
  hid_t file = H5Fcreate(path.c_str(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
  if (file < 0) return false;
  if (multi_read) {
    for (size_t r = 0; r < ids.size(); ++r) {
      hid_t g = H5Gcreate2(file, ("/read_" + ids[r]).c_str(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
      hid_t raw = H5Gcreate2(g, "Raw", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
      WriteStringAttr(raw, "read_id", ids[r]);
      WriteSignal(raw, raws[r]);
      H5Gclose(raw);
      hid_t ch = H5Gcreate2(g, "channel_id", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
      WriteChannel(ch, digitisation, range, offset);
      H5Gclose(ch);
      H5Gclose(g);
    }
  } else {
    hid_t raw = H5Gcreate2(file, "/Raw", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    hid_t reads = H5Gcreate2(raw, "Reads", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    for (size_t r = 0; r < ids.size(); ++r) {
      hid_t g = H5Gcreate2(reads, ("Read_" + std::to_string(r + 1)).c_str(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
      WriteStringAttr(g, "read_id", ids[r]);
      WriteSignal(g, raws[r]);
      H5Gclose(g);
    }
    H5Gclose(reads);
    H5Gclose(raw);
    hid_t ugk = H5Gcreate2(file, "/UniqueGlobalKey", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    hid_t ch = H5Gcreate2(ugk, "channel_id", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    WriteChannel(ch, digitisation, range, offset);
    H5Gclose(ch);
    H5Gclose(ugk);
  }
  return H5Fclose(file) >= 0;
}
}  // namespace sigmap

#else  // no HDF5 in this build

namespace sigmap {
bool SignalBatch::HaveFAST5() { return false; }
void SignalBatch::AddSignalsFromFAST5(const std::string &path) {
  ExitWithMessage("FAST5 input (" + path + ") needs a build with HDF5 (headers + libhdf5 were not found)");
}
bool SignalBatch::WriteFAST5(const std::string &, bool, const std::vector<std::string> &, float, float, float,
                             const std::vector<std::vector<int16_t>> &) {
  return false;
}
}  // namespace sigmap
#endif
