// see eh_jit.hpp
#include "eh_jit.hpp"

#include <hip/hiprtc.h>

#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {
// the kernel sources as the build saw them (Makefile: build/eh_sources.inc)
#include "eh_sources.inc"

std::string slot_val(unsigned s) {
    char b[32];
    if (s < EH_PROG_SLOT_FORC) snprintf(b, sizeof b, "par[%u]", s);
    else if (s < EH_PROG_SLOT_CONST) snprintf(b, sizeof b, "frc[%u]", s - EH_PROG_SLOT_FORC);
    else if (s < EH_PROG_SLOT_INSTR) snprintf(b, sizeof b, "c%u", s - EH_PROG_SLOT_CONST);
    else snprintf(b, sizeof b, "T.t[%u]", s - EH_PROG_SLOT_INSTR);
    return b;
}
std::string slot_adj(unsigned s) {
    char b[32];
    if (s < EH_PROG_SLOT_FORC) snprintf(b, sizeof b, "ap[%u]", s);
    else if (s < EH_PROG_SLOT_CONST) snprintf(b, sizeof b, "af[%u]", s - EH_PROG_SLOT_FORC);
    else if (s < EH_PROG_SLOT_INSTR) return "ac";
    else snprintf(b, sizeof b, "at[%u]", s - EH_PROG_SLOT_INSTR);
    return b;
}
// On-disk cache of compiled code objects: $EH_JIT_CACHE (a directory; "0" = off), else $XDG_CACHE_HOME/easyhybrid_hip, else
// ~/.cache/easyhybrid_hip.  Key = FNV-1a of everything that goes into the compilation.  File: u32 n, n x (u32 len, lowered
// name), u64 length + u64 FNV of the code object, code object.  Failures of any kind just mean "compile".
unsigned long long fnv(unsigned long long h, const void* p, size_t n) {
    const unsigned char* c = (const unsigned char*)p;
    for (size_t i = 0; i < n; ++i) { h ^= c[i]; h *= 1099511628211ull; }
    return h;
}
std::string cache_dir() {
    const char* e = getenv("EH_JIT_CACHE");
    if (e && !strcmp(e, "0")) return "";
    std::string d;
    if (e && *e) d = e;
    else if (const char* x = getenv("XDG_CACHE_HOME")) d = std::string(x) + "/easyhybrid_hip";
    else if (const char* h = getenv("HOME")) { d = std::string(h) + "/.cache"; (void)mkdir(d.c_str(), 0755); d += "/easyhybrid_hip"; }
    else return "";
    (void)mkdir(d.c_str(), 0755);
    return d;
}
bool cache_load(const std::string& path, int nnames, std::vector<std::string>* names, std::vector<char>* code) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    bool ok = false;
    unsigned n = 0;
    if (fread(&n, 4, 1, f) == 1 && (int)n == nnames) {
        ok = true;
        for (unsigned i = 0; i < n && ok; ++i) {
            unsigned len = 0;
            ok = fread(&len, 4, 1, f) == 1 && len < 4096;
            if (ok) { std::string nm(len, 0); ok = fread(&nm[0], 1, len, f) == len; names->push_back(nm); }
        }
        if (ok) {       // length + checksum of the code object: the HIP runtime crashes on a truncated ELF instead of refusing it
            unsigned long long len = 0, sum = 0;
            ok = fread(&len, 8, 1, f) == 1 && fread(&sum, 8, 1, f) == 1 && len > 0 && len < (64ull << 20);
            if (ok) { code->resize((size_t)len); ok = fread(code->data(), 1, code->size(), f) == code->size() && fnv(1469598103934665603ull, code->data(), code->size()) == sum; }
        }
    }
    fclose(f);
    return ok;
}
void cache_store(const std::string& path, const std::vector<std::string>& names, const std::vector<char>& code) {
    char tmp[64];
    snprintf(tmp, sizeof tmp, ".tmp%d", (int)getpid());
    const std::string t = path + tmp;
    FILE* f = fopen(t.c_str(), "wb");
    if (!f) return;
    const unsigned n = (unsigned)names.size();
    bool ok = fwrite(&n, 4, 1, f) == 1;
    for (auto& nm : names) { const unsigned len = (unsigned)nm.size(); ok = ok && fwrite(&len, 4, 1, f) == 1 && fwrite(nm.data(), 1, len, f) == len; }
    const unsigned long long len = code.size(), sum = fnv(1469598103934665603ull, code.data(), code.size());
    ok = ok && fwrite(&len, 8, 1, f) == 1 && fwrite(&sum, 8, 1, f) == 1 && fwrite(code.data(), 1, code.size(), f) == code.size();
    ok = (fclose(f) == 0) && ok;
    if (!ok || rename(t.c_str(), path.c_str()) != 0) (void)unlink(t.c_str());
}
// one forward statement / one block of adjoint updates for instruction word w; V / A map a slot to its value / adjoint expression
template <class V>
std::string fwd_expr(unsigned w, V val) {
    const std::string x = val((w >> 8) & 255u), y = val((w >> 16) & 255u), z = val(w >> 24);
    switch (w & 255u) {
        case EH_OP_ADD: return x + " + " + y;
        case EH_OP_SUB: return x + " - " + y;
        case EH_OP_MUL: return x + " * " + y;
        case EH_OP_DIV: return x + " / " + y;
        case EH_OP_NEG: return "-" + x;
        case EH_OP_EXP: return "__expf(" + x + ")";
        case EH_OP_LOG: return "__logf(" + x + ")";
        case EH_OP_POW: return "eh_pow(" + x + ", " + y + ")";
        case EH_OP_SQRT: return "sqrtf(" + x + ")";
        case EH_OP_TANH: return "eh_tanh(" + x + ")";
        case EH_OP_SIGMOID: return "eh_sigmoid(" + x + ")";
        case EH_OP_MAX: return "fmaxf(" + x + ", " + y + ")";
        case EH_OP_MIN: return "fminf(" + x + ", " + y + ")";
        case EH_OP_ABS: return "fabsf(" + x + ")";
        case EH_OP_SIN: return "sinf(" + x + ")";
        case EH_OP_COS: return "cosf(" + x + ")";
        case EH_OP_SELECT: return x + " > 0.0f ? " + y + " : " + z;
        case EH_OP_GT: return x + " > " + y + " ? 1.0f : 0.0f";
        default: return "0.0f";
    }
}
template <class V, class Aj>
std::string rev_stmts(unsigned w, V val, Aj adj) {
    const unsigned sa = (w >> 8) & 255u, sb = (w >> 16) & 255u, sc = w >> 24;
    const std::string x = val(sa), y = val(sb), A = adj(sa), B = adj(sb), C = adj(sc);
    switch (w & 255u) {
        case EH_OP_ADD: return A + " += g; " + B + " += g;";
        case EH_OP_SUB: return A + " += g; " + B + " += -g;";
        case EH_OP_MUL: return A + " += g * " + y + "; " + B + " += g * " + x + ";";
        case EH_OP_DIV: return A + " += g / " + y + "; " + B + " += -(g / " + y + ") * r;";
        case EH_OP_NEG: return A + " += -g;";
        case EH_OP_EXP: return A + " += g * r;";
        case EH_OP_LOG: return A + " += g / " + x + ";";
        case EH_OP_POW: return A + " += g * " + y + " * r / " + x + "; " + B + " += g * r * __logf(" + x + ");";
        case EH_OP_SQRT: return A + " += g * 0.5f / r;";
        case EH_OP_TANH: return A + " += g * (1.0f - r * r);";
        case EH_OP_SIGMOID: return A + " += g * r * (1.0f - r);";
        case EH_OP_MAX: return A + " += " + x + " >= " + y + " ? g : 0.0f; " + B + " += " + x + " >= " + y + " ? 0.0f : g;";
        case EH_OP_MIN: return A + " += " + x + " <= " + y + " ? g : 0.0f; " + B + " += " + x + " <= " + y + " ? 0.0f : g;";
        case EH_OP_ABS: return A + " += " + x + " > 0.0f ? g : (" + x + " < 0.0f ? -g : 0.0f);";
        case EH_OP_SIN: return A + " += g * cosf(" + x + ");";
        case EH_OP_COS: return A + " += -g * sinf(" + x + ");";
        case EH_OP_SELECT: return B + " += " + x + " > 0.0f ? g : 0.0f; " + C + " += " + x + " > 0.0f ? 0.0f : g;";
        default: return "";
    }
}
std::string const_decls(const float* c, int n) {
    std::string s;
    char b[128];
    for (int k = 0; k < n; ++k) {
        unsigned u;
        memcpy(&u, &c[k], 4);
        snprintf(b, sizeof b, "    const float c%d = __uint_as_float(0x%08xu);   // %.9g\n", k, u, (double)c[k]);
        s += b;
    }
    return s;
}
}   // namespace

static std::string eh_jit_loss_one(const EhLossProg1& lp, const char* fname) {
    const int n = (int)lp.code.size();
    auto val = [](unsigned sl) -> std::string {
        char b[32];
        if (sl == 0) return "yhat";
        if (sl == 1) return "yobs";
        if (sl < EH_PROG_SLOT_INSTR) snprintf(b, sizeof b, "c%u", sl - EH_PROG_SLOT_CONST);
        else snprintf(b, sizeof b, "t[%u]", sl - EH_PROG_SLOT_INSTR);
        return b;
    };
    auto adj = [](unsigned sl) -> std::string {
        char b[32];
        if (sl == 0) return "ayh";
        if (sl < EH_PROG_SLOT_INSTR) return "ac";
        snprintf(b, sizeof b, "at[%u]", sl - EH_PROG_SLOT_INSTR);
        return b;
    };
    char b[160];
    std::string s = std::string("__device__ __forceinline__ float ") + fname + "(float yhat, float yobs, float& dl) {\n" + const_decls(lp.consts.data(), (int)lp.consts.size());
    snprintf(b, sizeof b, "    float t[%d], at[%d] = {}, ayh = 0.0f, ac = 0.0f;\n", n, n);
    s += b;
    for (int i = 0; i < n; ++i) { snprintf(b, sizeof b, "    t[%d] = ", i); s += b + fwd_expr(lp.code[i], val) + ";\n"; }
    s += "    " + adj((unsigned)lp.out) + " += 1.0f;\n";
    for (int i = n - 1; i >= 0; --i) {
        snprintf(b, sizeof b, "    { const float g = at[%d], r = t[%d]; ", i, i);
        s += b + rev_stmts(lp.code[i], val, adj) + " (void)r; }\n";
    }
    s += "    dl = ayh; (void)ac;\n    return " + val((unsigned)lp.out) + ";\n}\n";
    return s;
}
// one function per target that has a program (its own, or the common one), and the dispatcher the kernels call with the (unrolled,
// hence compile-time) target index
std::string eh_jit_loss_source(const EhLossProg& lp) {
    std::string s;
    char b[96];
    for (int t = 0; t < 4; ++t)
        if (lp.has(t)) { snprintf(b, sizeof b, "eh_jit_loss_%d", t); s += eh_jit_loss_one(lp.of(t), b); }
    s += "__device__ __forceinline__ float eh_jit_loss(int target, float yhat, float yobs, float& dl) {\n    switch (target) {\n";
    for (int t = 0; t < 4; ++t)
        if (lp.has(t)) { snprintf(b, sizeof b, "        case %d: return eh_jit_loss_%d(yhat, yobs, dl);\n", t, t); s += b; }
    s += "        default: dl = 0.0f; return 0.0f;\n    }\n}\n";
    return s;
}

namespace {
}   // namespace

std::string eh_jit_mech_source(const eh_model_desc& d) {
    const int n = d.prog_len;
    std::string s;
    char b[256];
    snprintf(b, sizeof b, "struct EhJitTape { float t[%d]; };\n", n);
    s += b;
    std::string consts;
    for (int k = 0; k < d.prog_n_const; ++k) {
        unsigned u;
        memcpy(&u, &d.prog_const[k], 4);
        snprintf(b, sizeof b, "    const float c%d = __uint_as_float(0x%08xu);   // %.9g\n", k, u, (double)d.prog_const[k]);
        consts += b;
    }
    // ---- forward: the expressions of eh_prog_forward, one statement per instruction
    s += "__device__ __forceinline__ void eh_jit_fwd(const float* par, const float* frc, EhJitTape& T, float& y0, float& y1, float& y2) {\n" + consts;
    for (int i = 0; i < n; ++i) {
        const unsigned w = d.prog_code[i];
        const std::string x = slot_val((w >> 8) & 255u), y = slot_val((w >> 16) & 255u), z = slot_val(w >> 24);
        std::string e;
        switch (w & 255u) {
            case EH_OP_ADD: e = x + " + " + y; break;
            case EH_OP_SUB: e = x + " - " + y; break;
            case EH_OP_MUL: e = x + " * " + y; break;
            case EH_OP_DIV: e = x + " / " + y; break;
            case EH_OP_NEG: e = "-" + x; break;
            case EH_OP_EXP: e = "__expf(" + x + ")"; break;
            case EH_OP_LOG: e = "__logf(" + x + ")"; break;
            case EH_OP_POW: e = "eh_pow(" + x + ", " + y + ")"; break;
            case EH_OP_SQRT: e = "sqrtf(" + x + ")"; break;
            case EH_OP_TANH: e = "eh_tanh(" + x + ")"; break;
            case EH_OP_SIGMOID: e = "eh_sigmoid(" + x + ")"; break;
            case EH_OP_MAX: e = "fmaxf(" + x + ", " + y + ")"; break;
            case EH_OP_MIN: e = "fminf(" + x + ", " + y + ")"; break;
            case EH_OP_ABS: e = "fabsf(" + x + ")"; break;
            case EH_OP_SIN: e = "sinf(" + x + ")"; break;
            case EH_OP_COS: e = "cosf(" + x + ")"; break;
            case EH_OP_SELECT: e = x + " > 0.0f ? " + y + " : " + z; break;
            case EH_OP_GT: e = x + " > " + y + " ? 1.0f : 0.0f"; break;
            default: e = "0.0f"; break;
        }
        snprintf(b, sizeof b, "    T.t[%d] = ", i);
        s += b + e + ";\n";
    }
    for (int o = 0; o < EH_MAX_PROG_OUT; ++o) {
        snprintf(b, sizeof b, "    y%d = ", o);
        s += b + (o < d.prog_n_out ? slot_val((unsigned)d.prog_out[o]) : std::string("0.0f")) + ";\n";
    }
    s += "}\n";
    // ---- reverse sweep: the adjoint updates of eh_prog_reverse
    s += "__device__ __forceinline__ void eh_jit_rev(const float* par, const float* frc, const EhJitTape& T, float dy0, float dy1, float dy2, float* dp) {\n" + consts;
    snprintf(b, sizeof b, "    float ap[%d] = {}, af[%d] = {}, ac = 0.0f, at[%d] = {};\n", EH_MAX_PARAMS, EH_MAX_FORC, n);
    s += b;
    for (int o = 0; o < d.prog_n_out; ++o) {
        snprintf(b, sizeof b, " += dy%d;\n", o);
        s += "    " + slot_adj((unsigned)d.prog_out[o]) + b;
    }
    for (int i = n - 1; i >= 0; --i) {
        const unsigned w = d.prog_code[i], sa = (w >> 8) & 255u, sb = (w >> 16) & 255u, sc = w >> 24;
        const std::string x = slot_val(sa), y = slot_val(sb), A = slot_adj(sa), B = slot_adj(sb), C = slot_adj(sc);
        snprintf(b, sizeof b, "    { const float g = at[%d], r = T.t[%d]; ", i, i);
        s += b;
        switch (w & 255u) {
            case EH_OP_ADD: s += A + " += g; " + B + " += g;"; break;
            case EH_OP_SUB: s += A + " += g; " + B + " += -g;"; break;
            case EH_OP_MUL: s += A + " += g * " + y + "; " + B + " += g * " + x + ";"; break;
            case EH_OP_DIV: s += A + " += g / " + y + "; " + B + " += -(g / " + y + ") * r;"; break;
            case EH_OP_NEG: s += A + " += -g;"; break;
            case EH_OP_EXP: s += A + " += g * r;"; break;
            case EH_OP_LOG: s += A + " += g / " + x + ";"; break;
            case EH_OP_POW: s += A + " += g * " + y + " * r / " + x + "; " + B + " += g * r * __logf(" + x + ");"; break;
            case EH_OP_SQRT: s += A + " += g * 0.5f / r;"; break;
            case EH_OP_TANH: s += A + " += g * (1.0f - r * r);"; break;
            case EH_OP_SIGMOID: s += A + " += g * r * (1.0f - r);"; break;
            case EH_OP_MAX: s += A + " += " + x + " >= " + y + " ? g : 0.0f; " + B + " += " + x + " >= " + y + " ? 0.0f : g;"; break;
            case EH_OP_MIN: s += A + " += " + x + " <= " + y + " ? g : 0.0f; " + B + " += " + x + " <= " + y + " ? 0.0f : g;"; break;
            case EH_OP_ABS: s += A + " += " + x + " > 0.0f ? g : (" + x + " < 0.0f ? -g : 0.0f);"; break;
            case EH_OP_SIN: s += A + " += g * cosf(" + x + ");"; break;
            case EH_OP_COS: s += A + " += -g * sinf(" + x + ");"; break;
            case EH_OP_SELECT: s += B + " += " + x + " > 0.0f ? g : 0.0f; " + C + " += " + x + " > 0.0f ? 0.0f : g;"; break;
            default: break;
        }
        s += " (void)r; }\n";
    }
    snprintf(b, sizeof b, "    for (int j = 0; j < %d; ++j) dp[j] = ap[j];\n    (void)af; (void)ac;\n}\n", EH_MAX_PARAMS);
    s += b;
    return s;
}

// eh_row_act(layer, row) for EH_ACT_PER_NET: net k owns rows [sum_{j<k} net_hidden[j][l], + net_hidden[k][l]) of hidden layer l
// (the block placement of eh_create; `d` is the handle's normalised copy: net_depth filled in, net_hidden of an identity block =
// the net's last width); rows past the last net are padding (zero weights on both sides): identity
static std::string eh_jit_rowact_source(const eh_model_desc& d) {
    std::string s = "__device__ __forceinline__ int eh_row_act(int l, int row) {\n";
    char b[96];
    for (int l = 0; l < d.n_hidden; ++l) {
        snprintf(b, sizeof b, "    if (l == %d) return", l);
        s += b;
        if (d.n_nets == 0) {      // one network, an activation per hidden layer (`hidden_layers::Chain`): whatever the row
            snprintf(b, sizeof b, " %d;\n", d.net_activation[l]);
            s += b;
            continue;
        }
        int r0 = 0;
        for (int k = 0; k < d.n_nets; ++k) {
            r0 += d.net_hidden[k][l];
            snprintf(b, sizeof b, " row < %d ? %d :", r0, l < d.net_depth[k] ? d.net_activation[k] : (int)EH_ACT_IDENTITY);     // past its depth: identity block
            s += b;
        }
        snprintf(b, sizeof b, " %d;\n", (int)EH_ACT_IDENTITY);
        s += b;
    }
    snprintf(b, sizeof b, "    return %d;\n}\n", (int)EH_ACT_IDENTITY);
    s += b;
    return s;
}

// A build that does not get through the compiler is tried again more conservatively: level 1 without the SLP vectoriser (where the
// first attempt had it), level 2 at -O1.  Seen with the hiprtc / comgr that PyTorch bundles (ROCm 7.0) on the per-net-activation
// row-split kernels: "Illegal instruction detected: both data operands should be VGPR or AGPR" (a merged ds_write2_b32 with one
// operand in an accumulator register) at -O2 / -O3; the system compiler of ROCm 7.2 builds the same source.  A slower kernel beats
// none: these models have no kernel built ahead of time.
static thread_local int g_jit_level = 0;
bool eh_jit_build(const eh_model_desc& d, const EhArchInfo* A, int variant, int act, int fast, const EhNet* spec, bool with_p2p,
                  const EhLossProg* loss, EhJitKernel* out, std::string* log, bool allow_slp) {
    const EhVariant& V = A->var[variant];
    const bool prog = d.mech == EH_MECH_PROGRAM;
    const std::string mech = prog ? eh_jit_mech_source(d) : std::string();
    const std::string lsrc = loss ? eh_jit_loss_source(*loss) : std::string();
    const bool rowact = act == EH_ACT_PER_NET;
    const std::string rsrc = rowact ? eh_jit_rowact_source(d) : std::string();
    // (hiprtc has the HIP device runtime built in but no C library headers)
    std::string src = "typedef signed char int8_t; typedef unsigned char uint8_t; typedef int int32_t; typedef unsigned int uint32_t;\n"
                      "typedef long long int64_t; typedef unsigned long long uint64_t;\n";
    if (const char* dbg = getenv("EH_JIT_DEFINES")) {        // diagnostics: e.g. EH_JIT_DEFINES="EH_STAMPS EH_STAMPS_FINE" (tools/stamps.py)
        std::string d = dbg;
        size_t at = 0;
        while (at < d.size()) {
            size_t e = d.find(' ', at);
            if (e == std::string::npos) e = d.size();
            if (e > at) {                                     // NAME -> #define NAME 1 ; NAME=VALUE -> #define NAME VALUE
                std::string tok = d.substr(at, e - at);
                const size_t eq = tok.find('=');
                src += eq == std::string::npos ? "#define " + tok + " 1\n" : "#define " + tok.substr(0, eq) + " " + tok.substr(eq + 1) + "\n";
            }
            at = e + 1;
        }
    }
    if (prog) src += "#define EH_JIT_MECH 1\n";
    if (loss) src += "#define EH_JIT_LOSS 1\n";
    if (rowact) src += "#define EH_JIT_ROWACT 1\n";
    if (spec) {
        char b[512];
        snprintf(b, sizeof b, "#define EH_SPEC_NET %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %uu, %uu, %uu, %uu, %uu\n", spec->P, spec->K, spec->G, spec->T, spec->F,
                 spec->n_theta, spec->g_off, spec->scale_nn, spec->mech, spec->n_par, spec->loss, spec->n_out, spec->targ_out, spec->par_kind, spec->par_idx, spec->forc_col, spec->loss_t);
        src += b;
    }
    src += V.so ? "#include \"eh_bf16_sample.hpp\"\n" : V.bf16 ? "#include \"eh_wide_bf16.hpp\"\n" : A->wide ? "#include \"eh_wide.hpp\"\n" : "#include \"eh_device.hpp\"\n";
    const char* hnames[8] = {"eh_device.hpp", "eh_wide.hpp", "easyhybrid_hip.h", "eh_wide_bf16.hpp", "eh_bf16_sample.hpp", nullptr, nullptr, nullptr};
    const char* hsrc[8] = {eh_src_device, eh_src_wide, eh_src_public, eh_src_widebf, eh_src_bfs, nullptr, nullptr, nullptr};
    int nh = 5;
    if (prog) { hnames[nh] = "eh_jit_mech.inc"; hsrc[nh++] = mech.c_str(); }
    if (loss) { hnames[nh] = "eh_jit_loss.inc"; hsrc[nh++] = lsrc.c_str(); }
    if (rowact) { hnames[nh] = "eh_jit_rowact.inc"; hsrc[nh++] = rsrc.c_str(); }
    hiprtcProgram hp = nullptr;
    if (hiprtcCreateProgram(&hp, src.c_str(), "eh_jit.hip", nh, hsrc, hnames) != HIPRTC_SUCCESS) { *log = "hiprtcCreateProgram failed"; return false; }
    // which kernels: train + eval, the cross-GPU train kernel when asked for, and -- per-wave family, registry model with its descriptor
    // baked in, one target -- the multi-step train kernel (several one-workgroup steps per launch, eh_device.hpp EH_MODE_TRAIN_MULTI)
    int modes[4], nmode = 0;
    modes[nmode++] = EH_MODE_TRAIN; modes[nmode++] = EH_MODE_EVAL;
    if (with_p2p && !A->wide && !prog) modes[nmode++] = EH_MODE_TRAIN_P2P;
    if (!A->wide && !prog && !loss && !rowact && spec && spec->T == 1 && !(fast & 4)) modes[nmode++] = EH_MODE_TRAIN_MULTI;
    char name[4][160];
    for (int i = 0; i < nmode; ++i) {
        const int m = modes[i];
        if (A->wide && V.so && m == EH_MODE_TRAIN) snprintf(name[i], sizeof name[i], "eh_bfs_kernel<%d, %d, %d, %d, %d, %s, %d>", A->nbi, A->nbh, A->nl, V.nw, act, prog ? "true" : "false", V.bf16 == 2 ? 1 : 3);
        else if (A->wide && V.so) snprintf(name[i], sizeof name[i], "eh_widebf_kernel<%d, %d, %d, %d, %d, %d, %d, %s, %d>", A->nbi, A->nbh, A->nl, V.so, V.nw, act, m, prog ? "true" : "false", V.bf16 == 2 ? 1 : 3);
        else if (A->wide && V.bf16) snprintf(name[i], sizeof name[i], "eh_widebf_kernel<%d, %d, %d, %d, %d, %d, %d, %s, %d>", A->nbi, A->nbh, A->nl, V.nt, V.nw, act, m, prog ? "true" : "false", V.bf16 == 2 ? 1 : 3);
        else if (A->wide) snprintf(name[i], sizeof name[i], "eh_wide_kernel<%d, %d, %d, %d, %d, %d, %d, %s>", A->nbi, A->nbh, A->nl, V.nt, V.nw, act, m, prog ? "true" : "false");
        else snprintf(name[i], sizeof name[i], "eh_step_kernel<%d, %d, %d, %d, %d, %d, %d, %d>", A->nbi, A->nbh, A->nl, V.nt, V.nw, act, m,
                      (m == EH_MODE_EVAL) ? (fast & 5) : fast);       // (the eval kernels exist for FAST 0 / 1 / 4)
        hiprtcAddNameExpression(hp, name[i]);
    }
    // The flags of the Makefile, with one exception.  -fno-slp-vectorize is there because the SLP vectoriser miscompiled ONE group of
    // kernels (P <= 4 ReLU on shapes wider than one 16-row block: a loss sum lost in a packed accumulator); it
    // also costs the narrow nets 4 % -- the headline step 10.14 -> 9.73 us with the vectoriser on, A/B on one lease against the
    // round-1 library (9.66), profiles/r03/headline_ab.txt.  The one-block shapes (NBH = 1) never showed the bug -- round 1 ran its
    // whole suite and 3 600 fuzz configurations on them with the vectoriser on -- so their run-time kernels get it back; every
    // other shape, and everything built ahead of time, stays without.  EH_JIT_SLP=0 / 1 overrides (diagnostics).
    const char* const slp_env = getenv("EH_JIT_SLP");
    // (allow_slp = false -- the background build of "specialize" = 2, whose kernel takes over from the one built ahead of time at a
    //  timing-dependent step: same flags as that one, so the trajectory does not depend on when the switch happens; advisor, round 3)
    const bool slp_on = g_jit_level == 0 && allow_slp && (slp_env ? atoi(slp_env) != 0 : (A->nbh == 1 && !A->wide));
    std::vector<std::string> extra;                  // EH_JIT_EXTRA_OPTS="-mllvm -foo ..." (diagnostics): appended to the compile options
    if (const char* xo = getenv("EH_JIT_EXTRA_OPTS")) {
        std::string x = xo;
        size_t at = 0;
        while (at < x.size()) {
            size_t e = x.find(' ', at);
            if (e == std::string::npos) e = x.size();
            if (e > at) extra.push_back(x.substr(at, e - at));
            at = e + 1;
        }
    }
    std::vector<const char*> opts = {"--offload-arch=gfx950", g_jit_level >= 2 ? "-O1" : "-O3", slp_on ? "-fslp-vectorize" : "-fno-slp-vectorize", "-std=c++17"};
    for (const std::string& x : extra) opts.push_back(x.c_str());
    // ---- cached code object?
    std::string cpath;
    {
        const std::string dir = cache_dir();
        if (!dir.empty()) {
            unsigned long long h = 1469598103934665603ull;
            int ver[2] = {0, 0};
            hiprtcVersion(&ver[0], &ver[1]);
            h = fnv(h, ver, sizeof ver);
            h = fnv(h, src.data(), src.size()); h = fnv(h, mech.data(), mech.size()); h = fnv(h, lsrc.data(), lsrc.size()); h = fnv(h, rsrc.data(), rsrc.size());
            h = fnv(h, eh_src_device, sizeof eh_src_device); h = fnv(h, eh_src_wide, sizeof eh_src_wide); h = fnv(h, eh_src_public, sizeof eh_src_public);
            h = fnv(h, eh_src_widebf, sizeof eh_src_widebf); h = fnv(h, eh_src_bfs, sizeof eh_src_bfs);
            for (int m = 0; m < nmode; ++m) h = fnv(h, name[m], strlen(name[m]));
            for (const char* o : opts) h = fnv(h, o, strlen(o));
            char fn[64];
            snprintf(fn, sizeof fn, "/%016llx.eco", h);
            cpath = dir + fn;
        }
    }
    std::vector<std::string> lowered;
    std::vector<char> code;
    bool from_cache = !cpath.empty() && cache_load(cpath, nmode, &lowered, &code);
    if (getenv("EH_JIT_TRACE")) {          // diagnostics: what exactly is being compiled
        fprintf(stderr, "eh_jit: %s | slp %d | %s | cache %s%s\n", name[0], (int)slp_on, cpath.c_str(), from_cache ? "hit" : "miss", g_jit_level ? " | retry" : "");
        const size_t at = src.find("#define EH_SPEC_NET");
        if (at != std::string::npos) fprintf(stderr, "eh_jit: %s\n", src.substr(at, src.find('\n', at) - at).c_str());
    }
    if (!from_cache) {
        lowered.clear();
        const hiprtcResult rc = hiprtcCompileProgram(hp, (int)opts.size(), opts.data());
        size_t ls = 0;
        hiprtcGetProgramLogSize(hp, &ls);
        if (ls > 1) { log->resize(ls); hiprtcGetProgramLog(hp, &(*log)[0]); }
        if (rc != HIPRTC_SUCCESS) {
            *log = std::string("hiprtc: ") + hiprtcGetErrorString(rc) + "\n" + *log;
            hiprtcDestroyProgram(&hp);
            const int next = (g_jit_level == 0 && slp_on) ? 1 : (g_jit_level < 2 ? 2 : 3);
            if (next <= 2) {
                const std::string first = log->substr(0, 240);
                const int saved = g_jit_level;
                g_jit_level = next;
                const bool ok2 = eh_jit_build(d, A, variant, act, fast, spec, with_p2p, loss, out, log, allow_slp);
                g_jit_level = saved;
                if (ok2 && log->empty()) *log = std::string(next == 1 ? "(compiled with -fno-slp-vectorize" : "(compiled at -O1") + " after the first build failed: " + first + ")";
                return ok2;
            }
            return false;
        }
        size_t cs = 0;
        hiprtcGetCodeSize(hp, &cs);
        code.resize(cs);
        hiprtcGetCode(hp, code.data());
        bool names_ok = true;
        for (int m = 0; m < nmode; ++m) {
            const char* ln = nullptr;
            names_ok = names_ok && hiprtcGetLoweredName(hp, name[m], &ln) == HIPRTC_SUCCESS && ln;
            lowered.push_back(ln ? ln : "");
        }
        if (names_ok && !cpath.empty()) cache_store(cpath, lowered, code);
    }
    bool ok = hipModuleLoadData(&out->mod, code.data()) == hipSuccess;
    for (int i = 0; i < nmode && ok; ++i) {
        const int m = modes[i];
        ok = !lowered[i].empty() && hipModuleGetFunction(&out->fn[m], out->mod, lowered[i].c_str()) == hipSuccess;
        // (raises the dynamic-LDS limit where the runtime wants to be told; a refusal shows up as a failed launch, which the caller handles)
        if (ok) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(out->fn[m]), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          m == EH_MODE_TRAIN_MULTI ? (int)EH_LDS_LIMIT : (int)V.lds_bytes);
    }
    if (!ok && from_cache) (void)unlink(cpath.c_str());       // a stale or damaged entry: gone, the next build compiles
    (void)hipGetLastError();
    hiprtcDestroyProgram(&hp);
    if (!ok) { *log = "hipModuleLoadData / hipModuleGetFunction failed for the compiled program"; eh_jit_release(out); return false; }
    out->nw = V.nw;
    out->lds_bytes = V.lds_bytes;
    out->lds_eval_bytes = V.lds_eval;
    return true;
}

hipError_t eh_jit_launch(const EhJitKernel* k, int mode, int grid, hipStream_t stream, const EhNet* net, const EhStepArgs* args) {
    void* params[] = {const_cast<EhNet*>(net), const_cast<EhStepArgs*>(args)};
    if (mode < 0 || mode > 3 || !k->fn[mode]) return hipErrorNotSupported;
    if (mode == EH_MODE_TRAIN_MULTI) {      // one workgroup; its step-to-step state sits in LDS behind the work space
        const size_t lds_ms = k->lds_bytes + sizeof(float) * (size_t)eh_ms_extra_floats(net->n_theta, args->n_acc);
        if (grid != 1 || lds_ms > EH_LDS_LIMIT) return hipErrorInvalidValue;
        return hipModuleLaunchKernel(k->fn[mode], 1, 1, 1, 64u * (unsigned)k->nw, 1, 1, (unsigned)lds_ms, stream, params, nullptr);
    }
    const size_t lds = (mode == EH_MODE_EVAL && k->lds_eval_bytes) ? k->lds_eval_bytes : k->lds_bytes;
    return hipModuleLaunchKernel(k->fn[mode], (unsigned)grid, 1, 1, 64u * (unsigned)k->nw, 1, 1, (unsigned)lds, stream, params, nullptr);
}

void eh_jit_release(EhJitKernel* k) {
    if (k->mod) (void)hipModuleUnload(k->mod);
    k->mod = nullptr; k->fn[0] = k->fn[1] = k->fn[2] = k->fn[3] = nullptr;
}
