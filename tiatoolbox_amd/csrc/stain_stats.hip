// C entry points of the per-patch stain statistics (include/tiatoolbox_amd.h: tia_stain_stats_u8 and its workspace / path
// queries) and the dispatch between the three kernels -- see stain_stats_common.hpp for the map of the translation units.
// Reference: tools/stainextract.py:177-227,281-322, tools/stainnorm.py:49-66,81-85,103,
//            utils/misc.py:261-290,405-444, utils/transforms.py:209-231.
#include "stain_stats_common.hpp"

static size_t align256s(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" size_t tia_stain_stats_workspace_bytes(int64_t n, int64_t h, int64_t w) {
    if (n <= 0 || h <= 0 || w <= 0) return 0;
    return (size_t)n * (size_t)h * (size_t)w * 2u * sizeof(uint16_t);
}

extern "C" size_t tia_stain_stats_workspace_bytes_mode(int64_t n, int64_t h, int64_t w, int32_t mode) {
    if (n <= 0 || h <= 0 || w <= 0) return 0;
    // [bin cache][dictionary (Vahadane only)][per-patch redo flags of the register-resident kernel]
    size_t total = align256s(tia_stain_stats_workspace_bytes(n, h, w));
    const long hw = (long)h * (long)w;
    if (mode != TIA_MODE_VAHADANE && hw > tia::kBigImagePixels) {  // large images: state, candidate lists and bin codes of the
        const size_t big = align256s(tia::stain_stats_big_workspace_bytes(n, hw));  // multi-workgroup path (stain_stats_big.hip)
        total = total > big ? total : big;
    }
    if (mode == TIA_MODE_VAHADANE) total += align256s((size_t)n * (size_t)h * (size_t)w * sizeof(double2));
    return total + align256s((size_t)n * sizeof(int));
}

// patches of this shape (in whole 4-pixel groups, window selection on) go through the register-resident kernel: every size it
// can hold (16 groups per thread: up to 256 x 256).  Round 3 stopped at 224 x 224 because one patch in eight was handed back at
// 256 x 256 -- the window-placing sample was column-aligned there (sample_index) -- which made the pair slower than the streaming
// kernel alone; with the stratified sample nothing is handed back and both take the same time (2.14 ms per 4096 x 256^2,
// profiles/r04q_*), the register-resident one with a third of the HBM-side traffic.  select_mode 2 keeps the streaming kernel.
static bool stats_reg_shape(long hw, const tia_stain_params* params) {
    static const bool no_reg = tia::dev_env("TIA_STATS_NO_REG") != nullptr;  // developer switch (A/B measurements)
    const long reg_limit = tia::stain_stats_reg_pixel_limit();
    // (>= 4096 pixels: the sample of 4096 dwords is parked in the patch's 4 hw bytes of workspace)
    return !no_reg && params->mode != TIA_MODE_VAHADANE && params->select_mode == 0 && (hw & 3) == 0 && hw <= reg_limit &&
           hw >= tia::SAMPLE_TARGET;
}

extern "C" int tia_stain_stats_path(int64_t h, int64_t w, const tia_stain_params* params) {
    if (!params || h <= 0 || w <= 0) return TIA_EINVAL;
    return stats_reg_shape((long)h * (long)w, params) ? 1 : 0;
}

extern "C" int tia_stain_stats_u8(const uint8_t* d_img, int64_t n, int64_t h, int64_t w,
                                   const tia_stain_tables* d_tables, const tia_stain_params* params,
                                   double* d_stats, void* d_ws, size_t ws_bytes, void* stream) {
    if (!d_img || !d_tables || !params || !d_stats) return TIA_EINVAL;
    if (n <= 0 || h <= 0 || w <= 0) return TIA_EINVAL;
    if (params->mode < TIA_MODE_MACENKO || params->mode > TIA_MODE_GIVEN) return TIA_EINVAL;
    const long hw = (long)h * (long)w;
    if ((unsigned long long)hw * 3ull >= 0xffffffffull) return TIA_ESIZE;  // 32-bit histogram counts
    if (n > 0x7fffffffll) return TIA_ESIZE;
    const bool aligned = d_ws && (reinterpret_cast<uintptr_t>(d_ws) & 15) == 0;
    // the per-pixel bin cache is optional: without (enough) workspace the kernel recomputes values
    uint16_t* binws = (aligned && ws_bytes >= tia_stain_stats_workspace_bytes(n, h, w)) ? (uint16_t*)d_ws : nullptr;
    double2* dictws = nullptr;
    hipStream_t st = (hipStream_t)stream;
    if (params->mode == TIA_MODE_VAHADANE) {  // the dictionary scratch is not optional
        if (!aligned || ws_bytes < tia_stain_stats_workspace_bytes_mode(n, h, w, TIA_MODE_VAHADANE)) return TIA_ESIZE;
        if (params->dl_max_iter < 1 || !(params->dl_alpha >= 0.0)) return TIA_EINVAL;
        dictws = (double2*)((char*)d_ws + align256s(tia_stain_stats_workspace_bytes(n, h, w)));
        static const bool env_one = tia::dev_env("TIA_DL_ONE_KERNEL") != nullptr;  // developer switch (A/B measurements)
        if (params->dl_one_kernel || env_one)
            return tia::launch_stain_stats_stream(true, d_img, n, hw, d_tables, *params, d_stats, binws, dictws, nullptr, st);
        // the kernel pair: dictionary learning by replay (no dictionary traffic), then the common tail in MODE_VTAIL; what the first
        // kernel hands back (an unused atom's re-draw, more iterations than it records) goes through the one-kernel form
        int* redo = (int*)((char*)dictws + align256s((size_t)n * (size_t)hw * sizeof(double2)));
        if (hipMemsetAsync(redo, 0, (size_t)n * sizeof(int), st) != hipSuccess) return TIA_ELAUNCH;
        if (const int rc = tia::launch_vahadane_dl(d_img, n, hw, d_tables, *params, d_stats, redo, st); rc != TIA_OK) return rc;
        tia_stain_params tail = *params;
        tail.mode = tia::MODE_VTAIL;
        if (const int rc = tia::launch_stain_stats_stream(false, d_img, n, hw, d_tables, tail, d_stats, binws, nullptr, redo, st); rc != TIA_OK)
            return rc;
        return tia::launch_stain_stats_stream(true, d_img, n, hw, d_tables, *params, d_stats, binws, dictws, redo, st);
    }
    // LARGE images (above 4 x 256 x 256 pixels; Macenko / fixed / given matrices, default selection): every sweep over many
    // workgroups per image, decisions between sweeps in one-workgroup kernels (stain_stats_big.hip) -- the per-patch kernels below
    // give an image ONE workgroup.  The state lives at the front of the workspace (the bin cache is not used on this path).
    static const bool no_big = tia::dev_env("TIA_STATS_NO_BIG") != nullptr;  // developer switch: audit against the streaming kernel
    if (!no_big && hw > tia::kBigImagePixels && params->select_mode == 0 && aligned && ws_bytes >= tia::stain_stats_big_workspace_bytes(n, hw) &&
        n <= 65535)
        return tia::launch_stain_stats_big(d_img, n, hw, d_tables, *params, d_stats, d_ws, st);
    // Patches of up to 13 x 4096 pixels (224 x 224 and smaller) in whole 4-pixel groups go through the register-resident kernel
    // (the patch is read from HBM once); whatever it hands back -- and every other case -- through the streaming kernel.  Both
    // give the same bits.  Measured on MI355X (profiles/r03d_perf_stain*.txt): both kernels are bound by the vector ALU (about 150
    // instructions per pixel over the four sweeps, ~4 cycles each per wave), so the single read buys HBM traffic (1.25 GB instead of
    // 3.9 GB per 4096 x 224^2), not time; at 256 x 256 one 1024-thread workgroup per CU (16 register slots per thread) is slower than
    // two streaming workgroups that overlap each other's single-lane phases, so those patches stay on the streaming kernel.
    const bool reg_ok = stats_reg_shape(hw, params) &&
                        (reinterpret_cast<uintptr_t>(d_img) & 3) == 0 && aligned &&
                        ws_bytes >= tia_stain_stats_workspace_bytes_mode(n, h, w, params->mode);
    if (reg_ok) {
        int* redo = (int*)((char*)d_ws + align256s(tia_stain_stats_workspace_bytes(n, h, w)));
        if (hipMemsetAsync(redo, 0, (size_t)n * sizeof(int), st) != hipSuccess) return TIA_ELAUNCH;
        // (the register-resident kernel has no use for the bin cache: 16 KB per patch of it hold the window-placing sample)
        if (const int rc = tia::launch_stain_stats_reg(d_img, n, hw, d_tables, *params, d_stats, redo, reinterpret_cast<uint32_t*>(d_ws), st);
            rc != TIA_OK)
            return rc;
        return tia::launch_stain_stats_stream(false, d_img, n, hw, d_tables, *params, d_stats, binws, dictws, redo, st);
    }
    return tia::launch_stain_stats_stream(false, d_img, n, hw, d_tables, *params, d_stats, binws, dictws, nullptr, st);
}
