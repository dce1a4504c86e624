# EXPERIMENT OF ROUND 5, NOT APPLIED: XCD-weighted slices of the shared steps of fa_fwd16_w64 (run from csrc/ against the tree of commit e5e7119).
# Measured with the board's own per-XCD clocks as weights: 3.4-5.9 % SLOWER on cut launches (profiles/r5/xcd_balance_probe.txt): unequal slices
# stop coinciding with the two-parts-per-item structure, so workgroups get two partial segments (two prologues, two publishes / folds).

p='fa_fwd16_w64_kernel.inc'; s=open(p).read()
old='''    auto slice_bound = [&](uint32_t g) -> uint32_t {
        if constexpr (MASKT) { if (g >= Gs) return (uint32_t)total; }
        return (uint32_t)(total * g / Gs) - ((g & 1u) ? skew : 0u);
    };'''
new='''    // p.xbal (round 5): the eight XCDs of a part do not run at one clock under the power cap (measured inside this kernel: 1594 ... 1701 MHz
    // on one board, 1678 ... 1795 on another; the launch ends with its slowest XCD, 3 % after the average one: profiles/r5/wg_times_by_xcd.txt),
    // and workgroup numbers are XCD-contiguous (xcd_remap) -- so the launcher hands every XCD its share of the shared steps, p.xb[x] ...
    // p.xb[x + 1], sized by its measured clock, and the XCD's workgroups cut that range into equal slices.  The bounds are launch
    // parameters: the same for every launch of a process (one calibration per device), so results stay bitwise repeatable.
    auto slice_bound = [&](uint32_t g) -> uint32_t {
        if constexpr (MASKT) { if (g >= Gs) return (uint32_t)total; }
        if constexpr (!MASKT) {
            if (p.xbal) {
                const uint32_t q8 = G >> 3, x = g / q8;
                if (x >= 8u) return (uint32_t)total;
                return p.xb[x] + (uint32_t)((uint64_t)(p.xb[x + 1] - p.xb[x]) * (g - x * q8) / q8);
            }
        }
        return (uint32_t)(total * g / Gs) - ((g & 1u) ? skew : 0u);
    };'''
assert old in s; s=s.replace(old,new); open(p,'w').write(s)

p='fa_fwd16_w64.hip'; s=open(p).read()
old='''    const float* vsc;         // bf16pv16 kernels: 2^e of the V image's slabs'''
new='''    uint32_t xbal, xb[9];     // XCD-weighted slices of the shared steps (kernel: slice_bound; launcher: w64_xcd_bounds), xbal = 0: equal slices
    const float* vsc;         // bf16pv16 kernels: 2^e of the V image's slabs'''
assert old in s; s=s.replace(old,new)
old='''    uint32_t lazy;          // lazy reference mode (fp16 P thresholds; the fp8 variant has no lazy bodies and ignores it)
    uint32_t skew;
};'''
new='''    uint32_t lazy;          // lazy reference mode (fp16 P thresholds; the fp8 variant has no lazy bodies and ignores it)
    uint32_t skew;
    uint32_t xbal, xb[9];   // as W64Params
};'''
assert old in s; s=s.replace(old,new)
# host helper before launch_w64_kernel template
old='''template <typename KFN>
static hipError_t launch_w64_kernel(KFN kfn, const FwdParams& p, const W64Params& wp, hipStream_t stream) {'''
new='''// XCD-weighted slice bounds of the shared steps (kernel: slice_bound).  v[x] = relative clock of the XCD that runs workgroups with
// blockIdx % 8 == x (device_xcd_speeds: one calibration per device; all 1 = no information).  A workgroup of XCD x does W whole-round
// steps and s_x shared ones in (W + s_x) / v_x; equal finish times: s_x = (mean shared steps + W) v_x / mean(v) - W.
static bool w64_xcd_bounds(const FwdParams& p, uint32_t G, uint32_t n_items, uint32_t steps_per_item, uint32_t* xb) {
    if (p.causal || p.mask_kind == MK_BOOL || (G & 7u) || G != (uint32_t)w64_cu_count() || tuning().no_xcd_balance.load(std::memory_order_relaxed)) return false;
    const uint32_t full = n_items / G, rem = n_items % G;
    if (!rem) return false;
    float v[8];
    if (!device_xcd_speeds(v)) return false;
    const double total = (double)rem * steps_per_item, W = (double)full * steps_per_item, sbar = total / G;
    double mean = 0.0, s[8], sum = 0.0;
    for (int x = 0; x < 8; ++x) mean += v[x] / 8.0;
    for (int x = 0; x < 8; ++x) {
        s[x] = std::max((sbar + W) * v[x] / mean - W, 0.25 * sbar);  // (never less than a quarter of the equal share)
        sum += s[x];
    }
    double acc = 0.0;
    xb[0] = 0;
    for (int x = 0; x < 8; ++x) {
        acc += s[x] / sum;
        xb[x + 1] = x == 7 ? (uint32_t)total : (uint32_t)std::llround(total * acc);
    }
    return true;
}

template <typename KFN>
static hipError_t launch_w64_kernel(KFN kfn, const FwdParams& p, const W64Params& wp, hipStream_t stream) {'''
assert old in s; s=s.replace(old,new)
old='''    wp.skew = (uint32_t)tuning().w64_skew.load(std::memory_order_relaxed);
    // (skew = 0xffffffff would run'''
new='''    wp.skew = (uint32_t)tuning().w64_skew.load(std::memory_order_relaxed);
    wp.xbal = (wp.skew == 0 && w64_xcd_bounds(p, w64_grid(p), wp.n_items, w64_tiles_per_item(p), wp.xb)) ? 1u : 0u;
    // (skew = 0xffffffff would run'''
assert old in s; s=s.replace(old,new)
old='''    wp.skew = (uint32_t)tuning().w64_skew.load(std::memory_order_relaxed);
    const uint32_t grid = w64_grid(p);
    const size_t lds = 65536 + 4 * 32 * (512 + 16) + 16;'''
new='''    wp.skew = (uint32_t)tuning().w64_skew.load(std::memory_order_relaxed);
    const uint32_t grid = w64_grid(p);
    wp.xbal = (wp.skew == 0 && w64_xcd_bounds(p, grid, wp.n_items, wp.T, wp.xb)) ? 1u : 0u;
    const size_t lds = 65536 + 4 * 32 * (512 + 16) + 16;'''
assert old in s; s=s.replace(old,new)
open(p,'w').write(s)

p='kernels.h'; s=open(p).read()
old='''int device_cu_count();'''
new='''int device_cu_count();
// relative clocks of the current device's eight XCDs under load, indexed by blockIdx % 8 of a 1-D grid (mean 1); false: not known (not
// calibrated, a device without eight XCDs, balancing switched off).  tuning.hip: option "xcd_weights" sets them, "xcd_calibrate" measures them.
bool device_xcd_speeds(float (&v)[8]);'''
assert old in s; s=s.replace(old,new)
s=s.replace("quant_block_wg{0} /* tests / A-B: the block-wise quantiser in its one-workgroup-per-block form everywhere */;","quant_block_wg{0} /* tests / A-B: the block-wise quantiser in its one-workgroup-per-block form everywhere */,\n        no_xcd_balance{0} /* equal slices of the shared steps whatever the XCDs' clocks */;")
open(p,'w').write(s)

p='tuning.hip'; s=open(p).read()
old='''// ---- per-DEVICE launch state.'''
new='''// ---- relative XCD clocks per device (kernels.h device_xcd_speeds): set by the option "xcd_weights" = "v0,...,v7" (the CURRENT device)
static std::mutex g_xcd_mu;
static float g_xcd_v[64][8];
static bool g_xcd_known[64];

bool device_xcd_speeds(float (&v)[8]) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    std::lock_guard<std::mutex> lock(g_xcd_mu);
    if (!g_xcd_known[dev]) return false;
    for (int x = 0; x < 8; ++x) v[x] = g_xcd_v[dev][x];
    return true;
}

static bool set_xcd_weights(const char* value) {
    float v[8];
    int n = 0;
    const char* s = value;
    while (n < 8 && *s) {
        char* end = nullptr;
        const float f = strtof(s, &end);
        if (end == s || !(f > 0.5f && f < 2.0f)) return false;
        v[n++] = f;
        s = *end == ',' ? end + 1 : end;
    }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    std::lock_guard<std::mutex> lock(g_xcd_mu);
    if (n == 0) { g_xcd_known[dev] = false; return true; }  // "": forget
    if (n != 8) return false;
    for (int x = 0; x < 8; ++x) g_xcd_v[dev][x] = v[x];
    g_xcd_known[dev] = true;
    return true;
}

// ---- per-DEVICE launch state.'''
assert old in s; s=s.replace(old,new,1)
old='''    if (!strcmp(name, "softmax_tau") || !strcmp(name, "w64_tau")) {'''
new='''    if (!strcmp(name, "xcd_weights")) return set_xcd_weights(value);
    if (!strcmp(name, "softmax_tau") || !strcmp(name, "w64_tau")) {'''
assert old in s; s=s.replace(old,new,1)
s=s.replace('''        x->quant_block_wg.store(env_flag("UMFA_QUANT_BLOCK_WG"));''','''        x->quant_block_wg.store(env_flag("UMFA_QUANT_BLOCK_WG"));
        x->no_xcd_balance.store(env_flag("UMFA_NO_XCD_BALANCE"));''')
s=s.replace('''{"quant_block_wg", &t.quant_block_wg, true},''','''{"quant_block_wg", &t.quant_block_wg, true}, {"no_xcd_balance", &t.no_xcd_balance, true},''')
s=s.replace('''{"quant_block_wg", &t.quant_block_wg},''','''{"quant_block_wg", &t.quant_block_wg}, {"no_xcd_balance", &t.no_xcd_balance},''')
open(p,'w').write(s)
