// One-blob weight hand-over: the parameters of a network as ONE flat fp32 array in the order of the reference module's
// state_dict() (clairs/model.py:150-560; `num_batches_tracked` entries left out), instead of one cto_weights_add per tensor.
// This is what the torch custom ops (torch_ops.cpp: clairsto::cvt_forward / bigru_forward take `packed_weights`) and
// non-Python callers use; the manifest below is the single definition of that order, and tests check it name by name and
// size by size against the state_dict of the reference's own pickled modules (tests/golden/pickles.json.gz).
#include <cstring>
#include <string>
#include <utility>
#include <vector>
#include "common.h"

using namespace cto;

namespace {

using Manifest = std::vector<std::pair<std::string, int64_t>>;

void add_linear(Manifest& m, const std::string& p, int64_t out, int64_t in) {
    m.emplace_back(p + ".weight", out * in);
    m.emplace_back(p + ".bias", out);
}

// depth-wise conv -> BatchNorm -> point-wise conv (clairs/model.py:91-100), all without bias
void add_dwconv(Manifest& m, const std::string& p, int64_t c, int64_t out) {
    m.emplace_back(p + ".net.0.weight", c * 9);
    m.emplace_back(p + ".net.1.weight", c);
    m.emplace_back(p + ".net.1.bias", c);
    m.emplace_back(p + ".net.1.running_mean", c);
    m.emplace_back(p + ".net.1.running_var", c);
    m.emplace_back(p + ".net.2.weight", out * c);
}

void add_heads(Manifest& m, const char* const* names, int K) {
    for (int k = 0; k < K; ++k) add_linear(m, std::string(names[k]) + "_fc2", 128, 128);
    for (int k = 0; k < K; ++k) add_linear(m, std::string(names[k]) + "_fc3", 2, 128);
}

int manifest_cvt(const cto_cvt_cfg* cfg, Manifest& m) {
    CTO_REQUIRE(cfg && (cfg->n_out == 4 || cfg->n_out == 6), CTO_EINVAL, "CvT manifest: n_out must be 4 or 6");
    int64_t cin = CTO_NCHAN, w = CTO_NPOS;
    for (int s = 0; s < 3; ++s) {
        const int64_t C = cfg->emb_dim[s], inner = int64_t(64) * cfg->heads[s];
        CTO_REQUIRE(C > 0 && cfg->heads[s] > 0 && cfg->depth[s] > 0, CTO_EINVAL, "CvT manifest: stage %d config", s + 1);
        const std::string L = "layer" + std::to_string(s + 1);
        m.emplace_back(L + ".0.weight", C * cin * 9);
        m.emplace_back(L + ".0.bias", C);
        m.emplace_back(L + ".1.g", C);
        m.emplace_back(L + ".1.b", C);
        for (int d = 0; d < cfg->depth[s]; ++d) {
            const std::string P = L + ".2.layers." + std::to_string(d);
            m.emplace_back(P + ".0.norm.g", C);
            m.emplace_back(P + ".0.norm.b", C);
            add_dwconv(m, P + ".0.fn.to_q", C, inner);
            add_dwconv(m, P + ".0.fn.to_kv", C, 2 * inner);
            add_linear(m, P + ".0.fn.to_out.0", C, inner);
            m.emplace_back(P + ".1.norm.g", C);
            m.emplace_back(P + ".1.norm.b", C);
            add_linear(m, P + ".1.fn.net.0", 4 * C, C);
            add_linear(m, P + ".1.fn.net.3", C, 4 * C);
        }
        cin = C;
        w = (w + 1) / 2;
    }
    add_linear(m, "fc1", 128, cin * w);
    add_linear(m, "fc2", 2, 128);                 // in the state_dict, unused by forward (clairs/model.py:186)
    static const char* const names[6] = {"a", "c", "g", "t", "i", "d"};
    add_heads(m, names, cfg->n_out);
    return CTO_OK;
}

int manifest_bigru(int n_out, Manifest& m) {
    CTO_REQUIRE(n_out == 4 || n_out == 6, CTO_EINVAL, "BiGRU manifest: n_out must be 4 or 6");
    const struct { const char* name; int64_t in, h; } gru[2] = {{"lstm", CTO_NCHAN, 128}, {"lstm_2", 256, 192}};
    for (const auto& g : gru)
        for (const char* sfx : {"", "_reverse"}) {
            m.emplace_back(std::string(g.name) + ".weight_ih_l0" + sfx, 3 * g.h * g.in);
            m.emplace_back(std::string(g.name) + ".weight_hh_l0" + sfx, 3 * g.h * g.h);
            m.emplace_back(std::string(g.name) + ".bias_ih_l0" + sfx, 3 * g.h);
            m.emplace_back(std::string(g.name) + ".bias_hh_l0" + sfx, 3 * g.h);
        }
    add_linear(m, "fc1", 128, int64_t(CTO_NPOS) * 384);
    add_linear(m, "fc2", 128, 128);               // unused by forward (clairs/model.py:421)
    static const char* const names[6] = {"na", "nc", "ng", "nt", "ni", "nd"};
    add_heads(m, names, n_out);
    return CTO_OK;
}

int build(int kind, const cto_cvt_cfg* cfg, int n_out, Manifest& m) {
    CTO_REQUIRE(kind == 0 || kind == 1, CTO_EINVAL, "model kind must be 0 (CvT) or 1 (BiGRU)");
    return kind == 0 ? manifest_cvt(cfg, m) : manifest_bigru(n_out, m);
}

int create_packed(int kind, const float* packed, int64_t numel, const cto_cvt_cfg* cfg, int n_out, cto_model** out) {
    CTO_REQUIRE(packed && out && numel >= 0, CTO_EINVAL, "create_packed: null argument");
    Manifest m;
    int rc = build(kind, cfg, n_out, m);
    if (rc != CTO_OK) return rc;
    int64_t total = 0;
    for (const auto& e : m) total += e.second;
    CTO_REQUIRE(total == numel, CTO_EMISSING, "packed weights hold %lld values, the %s manifest needs %lld", (long long)numel,
                kind == 0 ? "CvT" : "BiGRU", (long long)total);
    cto_weights* w = cto_weights_new();
    int64_t off = 0;
    for (const auto& e : m) {
        if ((rc = cto_weights_add(w, e.first.c_str(), packed + off, e.second)) != CTO_OK) break;
        off += e.second;
    }
    if (rc == CTO_OK) rc = kind == 0 ? cto_cvt_create(w, cfg, out) : cto_bigru_create(w, n_out, out);
    cto_weights_free(w);
    return rc;
}

}  // namespace

extern "C" int64_t cto_model_manifest(int kind, const cto_cvt_cfg* cfg, int n_out, char* buf, size_t cap) try {
    Manifest m;
    const int rc = build(kind, cfg, n_out, m);
    if (rc != CTO_OK) return rc;
    std::string s;
    for (const auto& e : m) s += e.first + "\t" + std::to_string(e.second) + "\n";
    if (buf && cap > s.size()) memcpy(buf, s.c_str(), s.size() + 1);
    return int64_t(s.size()) + 1;
} CTO_CATCH("cto_model_manifest", int64_t)

extern "C" int cto_cvt_create_packed(const float* packed, int64_t numel, const cto_cvt_cfg* cfg, cto_model** out) try {
    return create_packed(0, packed, numel, cfg, cfg ? cfg->n_out : 0, out);
} CTO_CATCH("cto_cvt_create_packed", int)

extern "C" int cto_bigru_create_packed(const float* packed, int64_t numel, int n_out, cto_model** out) try {
    return create_packed(1, packed, numel, nullptr, n_out, out);
} CTO_CATCH("cto_bigru_create_packed", int)
