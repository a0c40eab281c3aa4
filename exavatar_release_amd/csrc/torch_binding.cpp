// Compiled autograd node of the drop-in surface: GaussianRasterizer.forward (reference avatar/common/nets/module.py:632-640)
// + its backward (triggered from avatar/main/train.py:46) for ONE render, bound to the C ABI of libexa_raster.so
// (include/exa_raster.h).  Host code only -- no kernels here, nothing of the library's internals: it fills the ABI's job
// structs from torch tensors and calls exa_raster_forward_batch / exa_raster_backward_batch, exactly what
// rasterizer._Rasterize does in Python.  Why it exists: the Python node costs the host ~140 us per forward and ~170 us per
// backward (Function.apply, _Job bookkeeping, 60 ctypes fields, the engine calling back into Python under the GIL) against
// ~140 us of DEVICE time for both on the C3 workload -- the surface ExAvatar calls was host-bound by 2.6x.  This node keeps
// the same semantics (capacity-mode render, zero-copy header report polled before the outputs leave forward, overflow
// repaired in place, same arena layouts => bit-identical results) at a fraction of the host time.
// Everything it does not cover (K > 1, constant prefixes, composites, stream capture, non-contiguous or non-float32 inputs,
// on_overflow='raise', debug probes) returns None and the caller takes the Python node.
#include <torch/extension.h>
#include <torch/csrc/autograd/custom_function.h>
#include <c10/hip/HIPStream.h>
#include <hip/hip_runtime_api.h>
#include <dlfcn.h>

#include <atomic>
#include <chrono>

#include "../../include/exa_raster.h"

namespace {

using at::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

struct Abi {
    int (*forward_batch)(const ExaRasterForwardJob*, int32_t, int32_t, void*) = nullptr;
    int (*forward_bin_batch)(const ExaRasterForwardJob*, int32_t, void*) = nullptr;
    int (*forward_render_batch)(const ExaRasterForwardJob*, int32_t, int32_t, void*) = nullptr;
    int (*read_header_async)(const void*, void*, void*) = nullptr;
    int (*backward_batch)(const ExaRasterBackwardJob*, int32_t, int32_t, void*) = nullptr;
    int (*workspace_sizes)(int32_t, int32_t, int32_t, uint64_t, ExaRasterWorkspaceSizes*) = nullptr;
    const char* (*last_error)(void) = nullptr;
    int (*version)(void) = nullptr;
    // header-report slots (16 bytes each) in pinned host memory, reserved for this node by rasterizer._HdrPool
    volatile uint32_t* slots_host = nullptr;
    uint64_t slots_dev = 0;
    int n_slots = 0;
    std::atomic<uint32_t> next{0};
    std::atomic<uint32_t> tag{1};
} g;

const char* g_declined = "";        // why the most recent call was not taken (diagnostics: _exa_torch.last_decline())
inline py::object decline(const char* why) {
    g_declined = why;
    return py::none();
}

void check_rc(int rc, const char* where) {
    if (rc == 0) return;
    const char* msg = g.last_error ? g.last_error() : nullptr;
    TORCH_CHECK(false, msg && *msg ? msg : "exa_raster error", " (status ", rc, ", ", where, ")");
}

inline bool f32c(const Tensor& t, const c10::Device& dev) {
    return t.defined() && t.scalar_type() == at::kFloat && t.device() == dev && t.is_contiguous();
}
inline bool f32c_opt(const c10::optional<Tensor>& t, const c10::Device& dev) { return !t.has_value() || f32c(*t, dev); }
inline const float* fptr(const c10::optional<Tensor>& t) { return t.has_value() ? t->data_ptr<float>() : nullptr; }
inline float* fptr_w(const Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }

// Everything of one render that is not a differentiable input; kept alive for the backward in a capsule.
struct State : torch::CustomClassHolder {
    ExaRasterSettings s{};
    Tensor bg, view, proj, campos;            // what s points at
    Tensor dens[3];                           // fused densification statistics (undefined = off)
    int32_t P = 0, sh_M = 0;
    uint64_t capacity = 0;
    int64_t need = 0, retried_from = 0;
    bool poison = false;
    Tensor ws, bins, radii, is_vis;           // ws: splat records | tile workspace | bin workspace (one arena); exact mode: the bin
                                              // workspace is allocated after the host has read the instance count -> `bins`
    uint64_t gb = 0, tb = 0;
    int edge[9] = {-1, -1, -1, -1, -1, -1, -1, -1, -1};      // argument position -> edge of the node (tensor arguments present only)
};

struct Outputs { Tensor color, depth, alpha; };

// Queue the forward of `st` (capacity mode), poll its header report, repair an overflow.  Returns the images.
Outputs run_forward(State& st, const Tensor& m3, const c10::optional<Tensor>& sh, const c10::optional<Tensor>& col,
                    const Tensor& op, const c10::optional<Tensor>& sc, const c10::optional<Tensor>& rot,
                    const c10::optional<Tensor>& cov, bool store_ctx, c10::hip::HIPStream stream) {
    const auto dev = m3.device();
    const int H = st.s.image_height, W = st.s.image_width;
    const auto u8 = at::TensorOptions().dtype(at::kByte).device(dev);
    Tensor planes = at::empty({5, H, W}, at::TensorOptions().dtype(at::kFloat).device(dev));
    st.radii = at::empty({st.P}, at::TensorOptions().dtype(at::kInt).device(dev));
    st.is_vis = at::empty({st.P}, at::TensorOptions().dtype(at::kBool).device(dev));
    ExaRasterForwardJob j{};
    j.settings = &st.s;
    j.P = st.P; j.sh_M = st.sh_M;
    j.means3D = m3.data_ptr<float>(); j.shs = fptr(sh); j.colors_precomp = fptr(col); j.opacities = op.data_ptr<float>();
    j.scales = fptr(sc); j.rotations = fptr(rot); j.cov3D_precomp = fptr(cov);
    j.radii = st.radii.data_ptr<int32_t>();
    j.is_vis = reinterpret_cast<uint8_t*>(st.is_vis.data_ptr<bool>());
    float* base = planes.data_ptr<float>();
    j.out_color = base; j.out_depth = base + 3 * (size_t)H * W; j.out_alpha = base + 4 * (size_t)H * W;
    j.keep_sorted_keys = 0;
    auto take_slot = [&](uint32_t& tag) {
        const uint32_t slot = g.next.fetch_add(1) % (uint32_t)g.n_slots;
        tag = g.tag.fetch_add(1);
        if (tag == 0) tag = g.tag.fetch_add(1);
        return slot;
    };
    if (st.capacity == 0) {
        // 'exact' (what upstream does): stage 1, read the instance count back (16-byte D2H copy + one stream synchronisation),
        // allocate exactly, stage 2
        ExaRasterWorkspaceSizes sz{};
        check_rc(g.workspace_sizes(st.P, W, H, 0, &sz), "workspace_sizes");
        st.gb = sz.geom_bytes; st.tb = sz.tile_bytes;
        st.ws = at::empty({(int64_t)(sz.geom_bytes + sz.tile_bytes)}, u8);
        if (st.poison) st.ws.fill_(255);
        uint8_t* w = st.ws.data_ptr<uint8_t>();
        j.geom_ws = w; j.tile_ws = w + sz.geom_bytes; j.bin_ws = nullptr; j.capacity = 0;
        j.host_header = nullptr; j.header_tag = 0;
        check_rc(g.forward_bin_batch(&j, 1, stream.stream()), "exa_raster_forward_bin_batch");
        uint32_t tag;
        volatile uint32_t* rep = g.slots_host + 4 * take_slot(tag);
        check_rc(g.read_header_async(j.tile_ws, const_cast<uint32_t*>(rep), stream.stream()), "exa_raster_read_header_async");
        stream.synchronize();
        st.need = rep[0];
        st.capacity = std::max<uint64_t>(rep[0], 64);          // (the header reports whole 64-instance batch slots)
        check_rc(g.workspace_sizes(st.P, W, H, st.capacity, &sz), "workspace_sizes");
        st.bins = at::empty({(int64_t)sz.bin_bytes}, u8);
        if (st.poison) st.bins.fill_(255);
        j.bin_ws = st.bins.data_ptr<uint8_t>(); j.capacity = st.capacity;
        check_rc(g.forward_render_batch(&j, 1, store_ctx ? 1 : 0, stream.stream()), "exa_raster_forward_render_batch");
    } else
    for (int attempt = 0; attempt < 2; ++attempt) {
        ExaRasterWorkspaceSizes sz{};
        check_rc(g.workspace_sizes(st.P, W, H, st.capacity, &sz), "workspace_sizes");
        st.gb = sz.geom_bytes; st.tb = sz.tile_bytes;
        st.ws = at::empty({(int64_t)(sz.geom_bytes + sz.tile_bytes + sz.bin_bytes)}, u8);
        if (st.poison) st.ws.fill_(255);
        uint8_t* w = st.ws.data_ptr<uint8_t>();
        j.geom_ws = w; j.tile_ws = w + sz.geom_bytes; j.bin_ws = w + sz.geom_bytes + sz.tile_bytes;
        j.capacity = st.capacity;
        uint32_t tag;
        const uint32_t slot = take_slot(tag);
        volatile uint32_t* rep = g.slots_host + 4 * slot;
        j.host_header = reinterpret_cast<void*>(g.slots_dev + 16ull * slot);
        j.header_tag = tag;
        check_rc(g.forward_batch(&j, 1, store_ctx ? 1 : 0, stream.stream()), "exa_raster_forward_batch");
        // The scatter stage stores the report ~35 us into the forward's kernels, about when the launches above have been queued:
        // spin on the word (no runtime call); a stream that is far behind is waited for.
        if (rep[3] != tag) {
            const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(2);
            while (rep[3] != tag && std::chrono::steady_clock::now() < t_end) {}
            if (rep[3] != tag) {
                stream.synchronize();
                TORCH_CHECK(rep[3] == tag, "exavatar_release_amd: the header report of a render never arrived");
            }
        }
        const uint32_t need = rep[0], overflow = rep[1];
        st.need = need;
        if (!overflow) break;
        TORCH_CHECK(attempt == 0, "exavatar_release_amd: a render overflowed the capacity its own report asked for (", need, ")");
        st.retried_from = (int64_t)st.capacity;
        st.capacity = ((uint64_t)std::max<uint32_t>(need, 64) + 63) / 64 * 64;      // exactly what the report names, as _rerender does
    }
    Outputs o;
    o.color = planes.narrow(0, 0, 3);
    o.depth = planes.narrow(0, 3, 1);
    o.alpha = planes.narrow(0, 4, 1);
    return o;
}

// argument positions of RasterizeFn::forward = positions of the gradients backward returns
enum { A_STATE, A_M3, A_M2, A_SH, A_COL, A_OP, A_SC, A_ROT, A_COV, A_COUNT };

struct RasterizeFn : public torch::autograd::Function<RasterizeFn> {
    static variable_list forward(AutogradContext* ctx, const c10::intrusive_ptr<State>& st, const Tensor& m3, const Tensor& m2,
                                 const c10::optional<Tensor>& sh, const c10::optional<Tensor>& col, const Tensor& op,
                                 const c10::optional<Tensor>& sc, const c10::optional<Tensor>& rot,
                                 const c10::optional<Tensor>& cov) {
        // AutogradContext::needs_input_grad counts the node's EDGES: the tensor arguments that are present, in argument order
        const bool present[A_COUNT] = {false, true, true, sh.has_value(), col.has_value(), true, sc.has_value(), rot.has_value(),
                                       cov.has_value()};
        for (int i = 0, e = 0; i < A_COUNT; ++i) st->edge[i] = present[i] ? e++ : -1;
        auto stream = c10::hip::getCurrentHIPStream(m3.device().index());
        Outputs o = run_forward(*st, m3, sh, col, op, sc, rot, cov, true, stream);
        ctx->saved_data["st"] = c10::IValue::make_capsule(st);
        ctx->save_for_backward({m3, sh.value_or(Tensor()), col.value_or(Tensor()), op, sc.value_or(Tensor()),
                                rot.value_or(Tensor()), cov.value_or(Tensor())});
        ctx->mark_non_differentiable({st->radii, st->is_vis});
        // outputs nobody differentiates (depth / alpha when the loss ignores them) reach backward undefined instead of as
        // freshly zero-filled 4 MB tensors: the kernels take a null pointer for "no gradient"
        ctx->set_materialize_grads(false);
        return {o.color, st->radii, o.depth, o.alpha, st->is_vis};
    }

    static Tensor grad_in(const Tensor& g, int64_t planes, int64_t H, int64_t W) {
        if (!g.defined()) return g;
        if (g.scalar_type() == at::kFloat && g.dim() == 3 && g.size(0) == planes && g.size(1) == H && g.size(2) == W &&
            g.is_contiguous())
            return g;
        return g.to(at::kFloat).expand({planes, H, W}).contiguous();
    }

    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        variable_list ret(A_COUNT);
        const Tensor &g_color_in = grads[0], &g_depth_in = grads[2], &g_alpha_in = grads[3];
        if (!g_color_in.defined() && !g_depth_in.defined() && !g_alpha_in.defined()) return ret;    // nothing reached the images
        auto st = c10::static_intrusive_pointer_cast<State>(ctx->saved_data["st"].toCapsule());
        auto needs = [&](int arg) { return st->edge[arg] >= 0 && ctx->needs_input_grad((size_t)st->edge[arg]); };
        auto saved = ctx->get_saved_variables();          // (checks the version counters of the inputs)
        const Tensor &m3 = saved[0], &sh = saved[1], &col = saved[2], &op = saved[3], &sc = saved[4], &rot = saved[5],
                     &cov = saved[6];
        const auto dev = m3.device();
        c10::DeviceGuard guard(dev);
        auto stream = c10::hip::getCurrentHIPStream(dev.index());
        const int64_t P = st->P, H = st->s.image_height, W = st->s.image_width;
        const auto f32 = at::TensorOptions().dtype(at::kFloat).device(dev);
        Tensor g_color = grad_in(g_color_in, 3, H, W);
        if (!g_color.defined()) g_color = at::zeros({3, H, W}, f32);
        Tensor g_depth = grad_in(g_depth_in, 1, H, W), g_alpha = grad_in(g_alpha_in, 1, H, W);
        const bool has_dens = st->dens[0].defined() || st->dens[1].defined() || st->dens[2].defined();
        // ONE arena for the small per-Gaussian gradients (layout of rasterizer._Rasterize.backward); dL/dsh stays its own tensor
        const bool want[7] = {needs(A_M3), needs(A_M2) || has_dens, needs(A_COL), needs(A_OP), needs(A_SC), needs(A_ROT), needs(A_COV)};
        static const int64_t width[7] = {3, 3, 3, 1, 3, 4, 6};
        int64_t total = 0;
        for (int i = 0; i < 7; ++i) total += want[i] ? width[i] : 0;
        Tensor d[7];
        if (total) {
            Tensor arena = at::empty({P * total}, f32);
            int64_t off = 0;
            for (int i = 0; i < 7; ++i)
                if (want[i]) {
                    d[i] = arena.narrow(0, off, P * width[i]).view({P, width[i]});
                    off += P * width[i];
                }
        }
        Tensor d_sh;
        if (needs(A_SH)) d_sh = at::empty({P, (int64_t)st->sh_M, 3}, f32);
        ExaRasterWorkspaceSizes sz{};
        check_rc(g.workspace_sizes((int32_t)P, (int32_t)W, (int32_t)H, st->capacity, &sz), "workspace_sizes");
        Tensor grad_ws = at::empty({(int64_t)sz.grad_bytes}, at::TensorOptions().dtype(at::kByte).device(dev));
        if (st->poison) grad_ws.fill_(255);
        ExaRasterBackwardJob b{};
        b.settings = &st->s;
        b.P = (int32_t)P; b.sh_M = st->sh_M;
        b.means3D = m3.data_ptr<float>(); b.shs = fptr_w(sh); b.colors_precomp = fptr_w(col); b.opacities = op.data_ptr<float>();
        b.scales = fptr_w(sc); b.rotations = fptr_w(rot); b.cov3D_precomp = fptr_w(cov);
        b.radii = st->radii.data_ptr<int32_t>();
        uint8_t* w = st->ws.data_ptr<uint8_t>();
        b.geom_ws = w; b.tile_ws = w + st->gb; b.capacity = st->capacity;
        b.bin_ws = st->bins.defined() ? st->bins.data_ptr<uint8_t>() : w + st->gb + st->tb;
        b.dL_dcolor = g_color.data_ptr<float>(); b.dL_ddepth = fptr_w(g_depth); b.dL_dalpha = fptr_w(g_alpha);
        b.grad_ws = grad_ws.data_ptr<uint8_t>();
        b.dL_dmeans3D = fptr_w(d[0]); b.dL_dmeans2D = fptr_w(d[1]); b.dL_dcolors = fptr_w(d[2]); b.dL_dopacity = fptr_w(d[3]);
        b.dL_dscales = fptr_w(d[4]); b.dL_drotations = fptr_w(d[5]); b.dL_dcov3D = fptr_w(d[6]); b.dL_dsh = fptr_w(d_sh);
        b.densify_grad_accum = fptr_w(st->dens[0]); b.densify_track_cnt = fptr_w(st->dens[1]);
        b.densify_radius_max = fptr_w(st->dens[2]);
        b.used_slots = st->need ? (uint32_t)((st->need + 63) / 64) : 0;     // one wave per batch slot IN USE
        check_rc(g.backward_batch(&b, 1, 0, stream.stream()), "exa_raster_backward_batch");
        ret[A_M3] = std::move(d[0]);
        if (needs(A_M2)) ret[A_M2] = std::move(d[1]);
        ret[A_SH] = std::move(d_sh);
        ret[A_COL] = std::move(d[2]);
        ret[A_OP] = std::move(d[3]);
        ret[A_SC] = std::move(d[4]);
        ret[A_ROT] = std::move(d[5]);
        ret[A_COV] = std::move(d[6]);
        return ret;
    }
};

// rasterize(settings tuple (the 12 fields of GaussianRasterizationSettings), means3D, means2D, shs, colors_precomp, opacities,
//           scales, rotations, cov3D_precomp, capacity (0 = 'exact': two stages with a host round trip, as upstream), auto_mode,
//           poison, densify_stats | None)
// -> None (not a call this node covers: take the Python node) or
//    (color, radii, depth, alpha, is_vis, needed instances, capacity that overflowed | 0)
py::object rasterize(const py::tuple& rs, const Tensor& m3, const Tensor& m2, const c10::optional<Tensor>& sh,
                     const c10::optional<Tensor>& col, const Tensor& op, const c10::optional<Tensor>& sc,
                     const c10::optional<Tensor>& rot, const c10::optional<Tensor>& cov, int64_t capacity, bool auto_mode,
                     bool poison, const py::object& densify) {
    if (!g.forward_batch || rs.size() != 12) return decline("not initialised, or the settings are not a 12-tuple");
    const auto dev = m3.device();
    if (!dev.is_cuda() || m3.dim() != 2 || m3.size(0) == 0 || capacity < 0) return decline("means3D is not a non-empty [P, 3] device tensor");
    if (!(f32c(m3, dev) && f32c(op, dev) && f32c_opt(sh, dev) && f32c_opt(col, dev) && f32c_opt(sc, dev) && f32c_opt(rot, dev) &&
          f32c_opt(cov, dev)))
        return decline("an input is not a contiguous float32 tensor on the device of means3D");
    if (!m2.defined() || m2.device() != dev) return decline("means2D is on another device");
    if (sh.has_value() == col.has_value()) return decline("shs / colors_precomp: not exactly one");                     // (the Python path raises upstream's messages)
    if ((sc.has_value() && rot.has_value()) == cov.has_value() || sc.has_value() != rot.has_value())
        return decline("scales + rotations / cov3D_precomp: not exactly one");
    const bool grad_on = at::GradMode::is_enabled();
    const bool need_ctx = grad_on && (m3.requires_grad() || m2.requires_grad() || op.requires_grad() ||
                                      (sh.has_value() && sh->requires_grad()) || (col.has_value() && col->requires_grad()) ||
                                      (sc.has_value() && sc->requires_grad()) || (rot.has_value() && rot->requires_grad()) ||
                                      (cov.has_value() && cov->requires_grad()));
    if (auto_mode && !need_ctx) capacity = 0;               // config.mode 'auto': a render nobody differentiates is sized exactly
    auto st = c10::make_intrusive<State>();
    // the settings tuple, field by field (module.py:609-622); the four tensors must already live on the device as float32
    for (int i : {4, 6, 7, 9})
        if (!THPVariable_Check(rs[i].ptr())) return decline("a camera field of the settings is not a tensor");
    st->bg = THPVariable_Unpack(rs[4].ptr());
    st->view = THPVariable_Unpack(rs[6].ptr());
    st->proj = THPVariable_Unpack(rs[7].ptr());
    st->campos = THPVariable_Unpack(rs[9].ptr());
    if (!(f32c(st->bg, dev) && f32c(st->view, dev) && f32c(st->proj, dev) && f32c(st->campos, dev)))
        return decline("a camera tensor of the settings is not contiguous float32 on the device");
    if (st->bg.numel() < 3 || st->view.numel() < 16 || st->proj.numel() < 16 || st->campos.numel() < 3)
        return decline("a camera tensor of the settings is too small");
    ExaRasterSettings& s = st->s;
    try {
        s.image_height = rs[0].cast<int32_t>(); s.image_width = rs[1].cast<int32_t>();
        s.tanfovx = rs[2].cast<float>(); s.tanfovy = rs[3].cast<float>();
        s.scale_modifier = rs[5].cast<float>();
        s.sh_degree = rs[8].cast<int32_t>();
        s.prefiltered = rs[10].cast<bool>() ? 1 : 0; s.debug = rs[11].cast<bool>() ? 1 : 0;
    } catch (const py::cast_error&) {
        return decline("a scalar field of the settings has an unexpected type");      // (the Python node converts it)
    }
    s.bg = st->bg.data_ptr<float>();
    s.viewmatrix = st->view.data_ptr<float>(); s.projmatrix = st->proj.data_ptr<float>();
    s.campos = st->campos.data_ptr<float>();
    st->P = (int32_t)m3.size(0);
    st->sh_M = sh.has_value() ? (int32_t)sh->size(1) : 0;
    st->capacity = ((uint64_t)capacity + 63) / 64 * 64;
    st->poison = poison;
    if (!densify.is_none()) {
        py::tuple dt = densify.cast<py::tuple>();
        if (dt.size() != 3) return decline("densify_stats is not a 3-tuple");
        for (int i = 0; i < 3; ++i)
            if (!dt[i].is_none()) st->dens[i] = THPVariable_Unpack(dt[i].ptr());
    }
    c10::DeviceGuard guard(dev);
    auto stream = c10::hip::getCurrentHIPStream(dev.index());
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream.stream(), &cap) != hipSuccess || cap != hipStreamCaptureStatusNone)
        return decline("the stream is being captured");
    g_declined = "";
    Tensor color, depth, alpha;
    if (need_ctx) {
        variable_list o = RasterizeFn::apply(st, m3, m2, sh, col, op, sc, rot, cov);
        color = o[0]; depth = o[2]; alpha = o[3];
    } else {
        at::NoGradGuard ng;
        Outputs o = run_forward(*st, m3, sh, col, op, sc, rot, cov, false, stream);
        color = o.color; depth = o.depth; alpha = o.alpha;
        st->ws = Tensor();
        st->bins = Tensor();
    }
    return py::make_tuple(color, st->radii, depth, alpha, st->is_vis, st->need, st->retried_from);
}

void init(const std::string& lib_path, uint64_t slots_host, uint64_t slots_dev, int n_slots) {
    void* h = dlopen(lib_path.c_str(), RTLD_NOW | RTLD_GLOBAL);        // (already mapped by ctypes: the same instance)
    TORCH_CHECK(h, "exavatar_release_amd: cannot open ", lib_path, ": ", dlerror());
    auto sym = [&](const char* name) {
        void* p = dlsym(h, name);
        TORCH_CHECK(p, "exavatar_release_amd: ", lib_path, " does not export ", name);
        return p;
    };
    g.version = reinterpret_cast<int (*)(void)>(sym("exa_raster_version"));
    TORCH_CHECK(g.version() == EXA_RASTER_VERSION, "exavatar_release_amd: the compiled autograd node was built against ABI ",
                EXA_RASTER_VERSION, ", the library is ", g.version(), " (rebuild: python -m exavatar_release_amd.build)");
    g.workspace_sizes = reinterpret_cast<decltype(g.workspace_sizes)>(sym("exa_raster_workspace_sizes"));
    g.last_error = reinterpret_cast<decltype(g.last_error)>(sym("exa_raster_last_error"));
    g.backward_batch = reinterpret_cast<decltype(g.backward_batch)>(sym("exa_raster_backward_batch"));
    g.forward_bin_batch = reinterpret_cast<decltype(g.forward_bin_batch)>(sym("exa_raster_forward_bin_batch"));
    g.forward_render_batch = reinterpret_cast<decltype(g.forward_render_batch)>(sym("exa_raster_forward_render_batch"));
    g.read_header_async = reinterpret_cast<decltype(g.read_header_async)>(sym("exa_raster_read_header_async"));
    g.slots_host = reinterpret_cast<volatile uint32_t*>(slots_host);
    g.slots_dev = slots_dev;
    g.n_slots = n_slots;
    TORCH_CHECK(slots_host && slots_dev && n_slots > 0, "exavatar_release_amd: the compiled autograd node needs header-report slots");
    g.forward_batch = reinterpret_cast<decltype(g.forward_batch)>(sym("exa_raster_forward_batch"));      // last: marks "ready"
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "compiled autograd node of exavatar_release_amd.GaussianRasterizer over the C ABI of libexa_raster.so";
    m.def("init", &init);
    m.def("rasterize", &rasterize);
    m.def("last_decline", []() { return std::string(g_declined); });
    m.attr("abi_version") = EXA_RASTER_VERSION;
}
