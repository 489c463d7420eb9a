// onnx_import.hip -- build() from the reference's own configuration (host code only).
//
// The reference's SuperPoint::build / SuperGlue::build (src/super_point.cpp:18-102, src/super_glue.cpp:21-147) try the cached
// engine first (deserialize_engine, :402-438 / :539-574), otherwise parse `onnx_file`, build, and write the cache
// (save_engine).  Same flow here: the cache is a URFW weight container, and "parse the ONNX file" is reading its initialisers
// -- ModelProto.graph (7) -> GraphProto.initializer (5) / .node (1) -> TensorProto dims (1), data_type (2), float_data (4),
// name (8), raw_data (9) -- with a protobuf wire-format reader of a hundred lines: no onnx / protobuf library.  The packing
// (layout of DESIGN.md section 3: SuperPoint [tap][cin][cout] + bias; SuperGlue [cin][cout] + bias with BatchNorm folded and
// the attention channels head-major) is the one of ur-mvo_amd/synth.py pack_sp / pack_sg and weights_io.py, float operation
// for float operation, so a file imported here and one imported by the Python tooling give the same container bytes
// (tests/test_abi_cpu.py).  Neither ONNX blob ships with the reference (.MISSING_LARGE_BLOBS): verified on files of the same
// structure written by weights_io.write_onnx only.
#include "../../include/urf.h"
#include "urf_common.h"

#include <math.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

namespace {

struct Tensor { std::vector<long> dims; std::vector<float> data; };
struct Node { std::string op; std::vector<std::string> in, out; };
struct Model { std::map<std::string, Tensor> inits; std::vector<Node> nodes; };

struct Span { const uint8_t *p; size_t n; };

bool varint(const Span &b, size_t &i, uint64_t &v) {
  v = 0;
  for (int s = 0; s < 64; s += 7) {
    if (i >= b.n) return false;
    const uint8_t c = b.p[i++];
    v |= (uint64_t)(c & 0x7F) << s;
    if (!(c & 0x80)) return true;
  }
  return false;
}

// one field of a message: number f, wire type wt, and either the varint value or the payload span
bool field(const Span &b, size_t &i, int &f, int &wt, uint64_t &v, Span &payload) {
  uint64_t key;
  if (!varint(b, i, key)) return false;
  f = (int)(key >> 3); wt = (int)(key & 7);
  payload = {nullptr, 0};
  if (wt == 0) return varint(b, i, v);
  size_t len = 0;
  if (wt == 1) len = 8;
  else if (wt == 5) len = 4;
  else if (wt == 2) { uint64_t l; if (!varint(b, i, l)) return false; len = (size_t)l; }
  else return false;
  if (len > b.n - i) return false;
  payload = {b.p + i, len};
  i += len;
  return true;
}

bool parse_tensor(const Span &b, std::string &name, Tensor &t, bool &is_float) {
  int dtype = 0;
  Span raw = {nullptr, 0};
  std::vector<float> floats;
  size_t i = 0;
  while (i < b.n) {
    int f, wt; uint64_t v; Span pl;
    if (!field(b, i, f, wt, v, pl)) return false;
    if (f == 1) {
      if (wt == 0) t.dims.push_back((long)v);
      else {                                   // packed repeated int64
        size_t j = 0; uint64_t d;
        while (j < pl.n) { if (!varint(pl, j, d)) return false; t.dims.push_back((long)d); }
      }
    } else if (f == 2) dtype = (int)v;
    else if (f == 4) {
      if (wt == 2) { const size_t k = pl.n / 4; const size_t o = floats.size(); floats.resize(o + k); memcpy(floats.data() + o, pl.p, k * 4); }
      else if (wt == 5) { float x; memcpy(&x, pl.p, 4); floats.push_back(x); }
    } else if (f == 8) name.assign((const char *)pl.p, pl.n);
    else if (f == 9) raw = pl;
  }
  is_float = dtype == 1;                       // FLOAT only (int64 shape constants etc. are not weights)
  if (!is_float) return true;
  if (raw.p) { t.data.resize(raw.n / 4); memcpy(t.data.data(), raw.p, t.data.size() * 4); }
  else t.data = floats;
  size_t want = 1;
  for (long d : t.dims) want *= (size_t)d;
  return t.data.size() == want;
}

int read_onnx(const char *path, Model &m, std::vector<uint8_t> &buf) {
  FILE *fp = fopen(path, "rb");
  URF_CHECK(fp, "cannot open ONNX file %s", path);
  fseek(fp, 0, SEEK_END);
  const long sz = ftell(fp);
  fseek(fp, 0, SEEK_SET);
  buf.resize(sz > 0 ? (size_t)sz : 0);
  const bool ok = sz > 0 && fread(buf.data(), 1, buf.size(), fp) == buf.size();
  fclose(fp);
  URF_CHECK(ok, "cannot read ONNX file %s", path);
  Span all = {buf.data(), buf.size()}, graph = {nullptr, 0};
  size_t i = 0;
  while (i < all.n) {
    int f, wt; uint64_t v; Span pl;
    URF_CHECK(field(all, i, f, wt, v, pl), "%s: malformed protobuf (model)", path);
    if (f == 7 && wt == 2) graph = pl;
  }
  URF_CHECK(graph.p, "%s: no graph in the ONNX model", path);
  i = 0;
  while (i < graph.n) {
    int f, wt; uint64_t v; Span pl;
    URF_CHECK(field(graph, i, f, wt, v, pl), "%s: malformed protobuf (graph)", path);
    if (f == 5 && wt == 2) {
      std::string name; Tensor t; bool is_float = false;
      URF_CHECK(parse_tensor(pl, name, t, is_float), "%s: malformed initialiser", path);
      if (is_float) m.inits[name] = std::move(t);
    } else if (f == 1 && wt == 2) {
      Node nd;
      size_t j = 0;
      while (j < pl.n) {
        int nf, nwt; uint64_t nv; Span np;
        URF_CHECK(field(pl, j, nf, nwt, nv, np), "%s: malformed node", path);
        if (nf == 1) nd.in.emplace_back((const char *)np.p, np.n);
        else if (nf == 2) nd.out.emplace_back((const char *)np.p, np.n);
        else if (nf == 4) nd.op.assign((const char *)np.p, np.n);
      }
      m.nodes.push_back(std::move(nd));
    }
  }
  return 0;
}

struct Conv { const Tensor *W; const Tensor *b; };   // b may be null (no bias input: zeros)

// (W, b) of every Conv node whose weight is an initialiser, first use only, graph order
std::vector<Conv> convs_in_order(const Model &m) {
  std::vector<Conv> out;
  std::vector<std::string> seen;
  for (const Node &nd : m.nodes) {
    if (nd.op != "Conv" || nd.in.size() < 2) continue;
    auto w = m.inits.find(nd.in[1]);
    if (w == m.inits.end()) continue;
    bool dup = false;
    for (const std::string &s : seen) dup = dup || s == nd.in[1];
    if (dup) continue;
    seen.push_back(nd.in[1]);
    const Tensor *b = nullptr;
    if (nd.in.size() > 2) { auto bi = m.inits.find(nd.in[2]); if (bi != m.inits.end()) b = &bi->second; }
    out.push_back({&w->second, b});
  }
  return out;
}

const Tensor *get(const Model &m, const std::string &name) {
  auto it = m.inits.find(name);
  return it == m.inits.end() ? nullptr : &it->second;
}

struct SpConv { const char *name; int cin, cout, k; };
const SpConv kSp[12] = {{"conv1a", 1, 64, 3},    {"conv1b", 64, 64, 3},   {"conv2a", 64, 64, 3},   {"conv2b", 64, 64, 3},
                        {"conv3a", 64, 128, 3},  {"conv3b", 128, 128, 3}, {"conv4a", 128, 128, 3}, {"conv4b", 128, 128, 3},
                        {"convPa", 128, 256, 3}, {"convPb", 256, 65, 1},  {"convDa", 128, 256, 3}, {"convDb", 256, 256, 1}};

int import_sp(const char *path, const Model &m, float *blob) {
  bool named = true;
  for (const SpConv &c : kSp) named = named && get(m, std::string(c.name) + ".weight") && get(m, std::string(c.name) + ".bias");
  std::vector<Conv> cv;
  if (named) {
    for (const SpConv &c : kSp) cv.push_back({get(m, std::string(c.name) + ".weight"), get(m, std::string(c.name) + ".bias")});
  } else {
    cv = convs_in_order(m);    // export order of superpoint/SP/model.py:58-86: conv1a .. conv4b, convPa, convPb, convDa, convDb
    URF_CHECK(cv.size() == 12, "%s: %zu Conv nodes with initialiser weights, SuperPoint has 12", path, cv.size());
  }
  float *o = blob;
  for (int i = 0; i < 12; ++i) {
    const SpConv &c = kSp[i];
    const Tensor &W = *cv[i].W;
    URF_CHECK(W.dims.size() == 4 && W.dims[0] == c.cout && W.dims[1] == c.cin && W.dims[2] == c.k && W.dims[3] == c.k,
              "%s: %s weight has the wrong shape", path, c.name);
    URF_CHECK(!cv[i].b || (long)cv[i].b->data.size() == c.cout, "%s: %s bias has the wrong shape", path, c.name);
    for (int ky = 0; ky < c.k; ++ky)             // OIHW -> [ky][kx][cin][cout]
      for (int kx = 0; kx < c.k; ++kx)
        for (int ci = 0; ci < c.cin; ++ci)
          for (int co = 0; co < c.cout; ++co) *o++ = W.data[(((size_t)co * c.cin + ci) * c.k + ky) * c.k + kx];
    for (int co = 0; co < c.cout; ++co) *o++ = cv[i].b ? cv[i].b->data[co] : 0.0f;
  }
  URF_CHECK((size_t)(o - blob) == URF_SP_BLOB_FLOATS, "internal: SuperPoint blob size");
  return 0;
}

// Conv1d weight [cout][cin] (a trailing kernel dimension of 1 allowed) + bias, with an optional BatchNorm folded in exactly like
// synth._fold_bn: s = gamma / sqrt(var + 1e-5) (f32), W' = W s, b' = (b - mean) s + beta (f32, no fused multiply-add)
struct Lin { std::vector<float> W, b; int cout, cin; };
int lin_from(const char *path, const Tensor *W, const Tensor *b, const Tensor *const bn[4], int cout, int cin, const char *what, Lin &out) {
  URF_CHECK(W && (W->dims.size() == 2 || (W->dims.size() == 3 && W->dims[2] == 1)) && W->dims[0] == cout && W->dims[1] == cin,
            "%s: %s weight is missing or has the wrong shape", path, what);
  URF_CHECK(!b || (long)b->data.size() == cout, "%s: %s bias has the wrong shape", path, what);
  out.cout = cout; out.cin = cin;
  out.W = W->data;
  out.b.assign(cout, 0.0f);
  if (b) out.b = b->data;
  if (bn) {
    for (int k = 0; k < 4; ++k) URF_CHECK(bn[k] && (long)bn[k]->data.size() == cout, "%s: BatchNorm of %s is missing or has the wrong shape", path, what);
    const float eps = 1e-5f;
    for (int o = 0; o < cout; ++o) {
      const float s = bn[0]->data[o] / sqrtf(bn[3]->data[o] + eps);
      for (int c = 0; c < cin; ++c) out.W[(size_t)o * cin + c] = out.W[(size_t)o * cin + c] * s;
      const float d = out.b[o] - bn[2]->data[o];
      const float e = d * s;
      out.b[o] = e + bn[1]->data[o];
    }
  }
  return 0;
}

// [cin][cout] + bias; rows (output channels) optionally taken in the order perm_out, columns (input channels) in perm_in
void put_lin(float *&o, const Lin &L, const int *perm_out, const int *perm_in) {
  for (int c = 0; c < L.cin; ++c)
    for (int r = 0; r < L.cout; ++r)
      *o++ = L.W[(size_t)(perm_out ? perm_out[r] : r) * L.cin + (perm_in ? perm_in[c] : c)];
  for (int r = 0; r < L.cout; ++r) *o++ = L.b[perm_out ? perm_out[r] : r];
}

int import_sg(const char *path, const Model &m, float *blob) {
  static const int kd[6] = {3, 32, 64, 128, 256, 256};
  static const int kenc_conv[5] = {0, 3, 6, 9, 12}, kenc_bn[4] = {1, 4, 7, 10};
  int perm[256];                                   // perm[c_new] = c_orig, c_new = h * 64 + d, c_orig = d * 4 + h
  for (int c = 0; c < 256; ++c) perm[c] = (c % 64) * 4 + c / 64;
  const bool named = get(m, "final_proj.weight") && get(m, "gnn.layers.0.attn.proj.0.weight") && get(m, "kenc.encoder.1.running_mean");
  std::vector<Lin> lins;                           // 5 keypoint-encoder layers, then per GNN layer q, k, v, merge, mlp.0, mlp.3, then final_proj
  float bin_score = 0.0f;
  auto bn_of = [&](const std::string &key, const Tensor *out[4]) {
    out[0] = get(m, key + ".weight"); out[1] = get(m, key + ".bias"); out[2] = get(m, key + ".running_mean"); out[3] = get(m, key + ".running_var");
  };
  if (named) {
    for (int i = 0; i < 5; ++i) {
      const std::string k = "kenc.encoder." + std::to_string(kenc_conv[i]);
      const Tensor *bn[4];
      if (i < 4) bn_of("kenc.encoder." + std::to_string(kenc_bn[i]), bn);
      Lin L;
      if (lin_from(path, get(m, k + ".weight"), get(m, k + ".bias"), i < 4 ? bn : nullptr, kd[i + 1], kd[i], k.c_str(), L)) return -2;
      lins.push_back(std::move(L));
    }
    for (int l = 0; l < 18; ++l) {
      const std::string p = "gnn.layers." + std::to_string(l);
      const std::string keys[6] = {p + ".attn.proj.0", p + ".attn.proj.1", p + ".attn.proj.2", p + ".attn.merge", p + ".mlp.0", p + ".mlp.3"};
      const int co[6] = {256, 256, 256, 256, 512, 256}, ci[6] = {256, 256, 256, 256, 512, 512};
      for (int j = 0; j < 6; ++j) {
        const Tensor *bn[4];
        if (j == 4) bn_of(p + ".mlp.1", bn);
        Lin L;
        if (lin_from(path, get(m, keys[j] + ".weight"), get(m, keys[j] + ".bias"), j == 4 ? bn : nullptr, co[j], ci[j], keys[j].c_str(), L)) return -2;
        lins.push_back(std::move(L));
      }
    }
    Lin L;
    if (lin_from(path, get(m, "final_proj.weight"), get(m, "final_proj.bias"), nullptr, 256, 256, "final_proj", L)) return -2;
    lins.push_back(std::move(L));
    const Tensor *bs = get(m, "bin_score");
    URF_CHECK(bs && bs->data.size() == 1, "%s: bin_score is missing", path);
    bin_score = bs->data[0];
  } else {
    // constant folding renames the Conv weights and folds BatchNorm into them: the distinct Conv weights in graph order.  A
    // graph that still carries BatchNormalization nodes has the same Conv count and would silently lose them.
    int nbn = 0;
    for (const Node &nd : m.nodes) nbn += nd.op == "BatchNormalization";
    URF_CHECK(nbn == 0, "%s: %d BatchNormalization nodes with renamed initialisers: export with constant folding (BatchNorm "
              "folded into the Conv weights) or keep the parameter names", path, nbn);
    const std::vector<Conv> cv = convs_in_order(m);
    URF_CHECK(cv.size() == 5 + 6 * 18 + 1, "%s: %zu distinct Conv weights, SuperGlue has %d", path, cv.size(), 5 + 6 * 18 + 1);
    size_t q = 0;
    for (int i = 0; i < 5; ++i, ++q) {
      Lin L;
      if (lin_from(path, cv[q].W, cv[q].b, nullptr, kd[i + 1], kd[i], "keypoint encoder", L)) return -2;
      lins.push_back(std::move(L));
    }
    for (int l = 0; l < 18; ++l) {
      const int co[6] = {256, 256, 256, 256, 512, 256}, ci[6] = {256, 256, 256, 256, 512, 512};
      for (int j = 0; j < 6; ++j, ++q) {
        Lin L;
        if (lin_from(path, cv[q].W, cv[q].b, nullptr, co[j], ci[j], "GNN layer", L)) return -2;
        lins.push_back(std::move(L));
      }
    }
    Lin L;
    if (lin_from(path, cv[q].W, cv[q].b, nullptr, 256, 256, "final_proj", L)) return -2;
    lins.push_back(std::move(L));
    std::vector<float> scal_named, scal_any;
    for (const auto &kv : m.inits)
      if (kv.second.data.size() == 1) { scal_any.push_back(kv.second.data[0]); if (kv.first.find("bin_score") != std::string::npos) scal_named.push_back(kv.second.data[0]); }
    const std::vector<float> &sc = scal_named.empty() ? scal_any : scal_named;
    URF_CHECK(sc.size() == 1, "%s: cannot identify bin_score (%zu scalar initialisers)", path, sc.size());
    bin_score = sc[0];
  }
  float *o = blob;
  size_t q = 0;
  for (int i = 0; i < 5; ++i) put_lin(o, lins[q++], nullptr, nullptr);
  for (int l = 0; l < 18; ++l) {
    for (int j = 0; j < 3; ++j) put_lin(o, lins[q++], perm, nullptr);   // q, k, v: output channels head-major
    put_lin(o, lins[q++], nullptr, perm);                               // merge: input channels head-major
    put_lin(o, lins[q++], nullptr, nullptr);                            // mlp.0 (BatchNorm folded)
    put_lin(o, lins[q++], nullptr, nullptr);                            // mlp.3
  }
  put_lin(o, lins[q++], nullptr, nullptr);
  *o++ = bin_score;
  URF_CHECK((size_t)(o - blob) == URF_SG_BLOB_FLOATS, "internal: SuperGlue blob size %zu", (size_t)(o - blob));
  return 0;
}

bool file_exists(const char *p) {
  if (!p || !*p) return false;
  FILE *f = fopen(p, "rb");
  if (!f) return false;
  fclose(f);
  return true;
}

}  // namespace

extern "C" int urf_onnx_import(const char *onnx_file, int kind, float *blob, size_t n_floats) {
  URF_CHECK(onnx_file && blob && (kind == 1 || kind == 2), "urf_onnx_import: bad argument");
  URF_CHECK(n_floats == (kind == 1 ? (size_t)URF_SP_BLOB_FLOATS : (size_t)URF_SG_BLOB_FLOATS), "urf_onnx_import: the blob of kind %d holds %d floats",
            kind, kind == 1 ? URF_SP_BLOB_FLOATS : URF_SG_BLOB_FLOATS);
  Model m;
  std::vector<uint8_t> buf;
  if (read_onnx(onnx_file, m, buf)) return -2;
  return kind == 1 ? import_sp(onnx_file, m, blob) : import_sg(onnx_file, m, blob);
}

// deserialize_engine() -- else build from onnx_file -- then save_engine(): src/super_point.cpp:21-32,99-101 / src/super_glue.cpp:21-33,145-146
template <typename H, typename BuildFile, typename Build>
static int build_config(H *h, const char *engine_file, const char *onnx_file, int kind, BuildFile build_file, Build build) {
  URF_CHECK(h, "build: null handle");
  if (file_exists(engine_file)) return build_file(h, engine_file);
  URF_CHECK(onnx_file && *onnx_file, "build: engine_file '%s' does not exist and no onnx_file is configured", engine_file ? engine_file : "");
  std::vector<float> blob(kind == 1 ? URF_SP_BLOB_FLOATS : URF_SG_BLOB_FLOATS);
  if (urf_onnx_import(onnx_file, kind, blob.data(), blob.size())) return -2;
  if (build(h, blob.data(), blob.size())) return -1;
  if (engine_file && *engine_file && urf_weights_save(engine_file, kind, blob.data(), blob.size()))
    fprintf(stderr, "liburf_front: built from %s, but the engine cache %s could not be written (%s)\n", onnx_file, engine_file, urf_last_error());
  return 0;
}

extern "C" int urf_sp_build_config(urf_sp *h, const char *engine_file, const char *onnx_file) {
  return build_config(h, engine_file, onnx_file, 1, urf_sp_build_file, urf_sp_build);
}
extern "C" int urf_pm_build_config(urf_pm *h, const char *engine_file, const char *onnx_file) {
  return build_config(h, engine_file, onnx_file, 2, urf_pm_build_file, urf_pm_build);
}
