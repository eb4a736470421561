// Thin pybind11 binding of the C ABI in include/gsr.h -- the form north_star words ("thin C++/pybind11 C-ABI
// extension") and the shape upstream's own binding has: the extension only marshals pointers and sizes, every buffer is
// allocated by the caller (a Python callable handed through as the allocator, exactly like upstream's resizeFunctional
// lambdas hand torch tensors' resize_ through).  Same library, same entry points as the ctypes binding in
// gaussmart_amd/_lib.py; tests/test_pybind_binding.py checks both produce identical bits.
// What it replaces in the reference: the `_C` module of submodules/diff-surfel-rasterization
// (rasterize_gaussians / rasterize_gaussians_backward; call site gaussian_renderer/__init__.py:97-106).
// No torch headers: device pointers cross as integers (tensor.data_ptr()), the stream as an integer handle.
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>
#include <stdexcept>
#include <string>
#include "../../include/gsr.h"

namespace py = pybind11;

namespace {
template <typename T>
T* ptr(uintptr_t v) { return reinterpret_cast<T*>(v); }

struct AllocCtx { py::object fn; std::string error; };

// the library calls this synchronously from inside gsr_forward / gsr_backward, on the calling thread (GIL held)
void* alloc_trampoline(void* ctx, int32_t which, size_t bytes) {
    auto* a = static_cast<AllocCtx*>(ctx);
    try {
        py::object r = a->fn(which, bytes);
        return r.is_none() ? nullptr : reinterpret_cast<void*>(r.cast<uintptr_t>());
    } catch (const std::exception& e) {
        a->error = e.what();
        return nullptr;
    }
}

GsrView make_view(const py::dict& v) {
    GsrView o{};
    o.width = v["width"].cast<int32_t>(); o.height = v["height"].cast<int32_t>();
    o.tanfovx = v["tanfovx"].cast<float>(); o.tanfovy = v["tanfovy"].cast<float>();
    o.scale_modifier = v["scale_modifier"].cast<float>();
    o.sh_degree = v["sh_degree"].cast<int32_t>(); o.sh_coeffs = v["sh_coeffs"].cast<int32_t>();
    o.channels = v["channels"].cast<int32_t>(); o.flags = v["flags"].cast<uint32_t>();
    o.bg = ptr<const float>(v["bg"].cast<uintptr_t>());
    o.viewmatrix = ptr<const float>(v["viewmatrix"].cast<uintptr_t>());
    o.projmatrix = ptr<const float>(v["projmatrix"].cast<uintptr_t>());
    o.campos = ptr<const float>(v["campos"].cast<uintptr_t>());
    return o;
}

uintptr_t opt(const py::dict& d, const char* k) { return d.contains(k) && !d[k].is_none() ? d[k].cast<uintptr_t>() : 0; }

GsrGaussians make_gaussians(const py::dict& g) {
    GsrGaussians o{};
    o.count = g["count"].cast<int32_t>();
    o.means3D = ptr<const float>(opt(g, "means3D")); o.shs = ptr<const float>(opt(g, "shs"));
    o.colors_precomp = ptr<const float>(opt(g, "colors_precomp")); o.opacities = ptr<const float>(opt(g, "opacities"));
    o.scales = ptr<const float>(opt(g, "scales")); o.rotations = ptr<const float>(opt(g, "rotations"));
    o.transmat_precomp = ptr<const float>(opt(g, "transmat_precomp")); o.shs_rest = ptr<const float>(opt(g, "shs_rest"));
    return o;
}

void check(int32_t rc, const AllocCtx& a) {
    if (rc == GSR_OK) return;
    if (!a.error.empty()) throw std::runtime_error("allocator callback failed: " + a.error);
    throw std::runtime_error(std::string("libgsr_hip: ") + gsr_last_error() + " (code " + std::to_string(rc) + ")");
}
}  // namespace

PYBIND11_MODULE(_gsr_pybind, m) {
    m.doc() = "pybind11 binding of libgsr_hip.so (include/gsr.h): pointer / size marshalling only";
    m.def("abi_version", [] { return gsr_abi_version(); });
    m.def("last_error", [] { return std::string(gsr_last_error()); });
    // forward(view, gaussians, out_color, out_allmap, radii, alloc(which, bytes) -> device pointer, stream)
    //   -> (num_rendered, geom, binning, image): the buffers the allocator returned, to be kept for the backward
    m.def("forward", [](const py::dict& view, const py::dict& gaussians, uintptr_t out_color, uintptr_t out_allmap,
                        uintptr_t radii, py::object alloc, uintptr_t stream) {
        GsrView v = make_view(view);
        GsrGaussians g = make_gaussians(gaussians);
        GsrForwardOut out{};
        out.out_color = ptr<float>(out_color); out.out_allmap = ptr<float>(out_allmap); out.radii = ptr<int32_t>(radii);
        AllocCtx ctx{std::move(alloc), {}};
        const int32_t rc = gsr_forward(&v, &g, &out, alloc_trampoline, &ctx, reinterpret_cast<gsr_stream_t>(stream));
        check(rc, ctx);
        return py::make_tuple(out.num_rendered, reinterpret_cast<uintptr_t>(out.geom), reinterpret_cast<uintptr_t>(out.binning),
                              reinterpret_cast<uintptr_t>(out.image));
    }, py::arg("view"), py::arg("gaussians"), py::arg("out_color"), py::arg("out_allmap"), py::arg("radii"), py::arg("alloc"),
       py::arg("stream"));
    // backward(view, gaussians, num_rendered, radii, geom, binning, image, dL_dcolor, dL_dallmap, grads, alloc, stream)
    //   grads: dict of device pointers by GsrGrads field name (absent / None = NULL)
    m.def("backward", [](const py::dict& view, const py::dict& gaussians, int32_t num_rendered, uintptr_t radii, uintptr_t geom,
                         uintptr_t binning, uintptr_t image, uintptr_t dL_dcolor, uintptr_t dL_dallmap, const py::dict& grads,
                         py::object alloc, uintptr_t stream) {
        GsrView v = make_view(view);
        GsrGaussians g = make_gaussians(gaussians);
        GsrGrads gr{};
        gr.dL_dmeans3D = ptr<float>(opt(grads, "dL_dmeans3D")); gr.dL_dmeans2D = ptr<float>(opt(grads, "dL_dmeans2D"));
        gr.dL_dopacity = ptr<float>(opt(grads, "dL_dopacity")); gr.dL_dshs = ptr<float>(opt(grads, "dL_dshs"));
        gr.dL_dcolors = ptr<float>(opt(grads, "dL_dcolors")); gr.dL_dscales = ptr<float>(opt(grads, "dL_dscales"));
        gr.dL_drotations = ptr<float>(opt(grads, "dL_drotations")); gr.dL_dtransmat = ptr<float>(opt(grads, "dL_dtransmat"));
        gr.dL_dshs_rest = ptr<float>(opt(grads, "dL_dshs_rest"));
        AllocCtx ctx{std::move(alloc), {}};
        const int32_t rc = gsr_backward(&v, &g, num_rendered, ptr<const int32_t>(radii), ptr<const void>(geom),
                                        ptr<const void>(binning), ptr<const void>(image), ptr<const float>(dL_dcolor),
                                        ptr<const float>(dL_dallmap), &gr, alloc_trampoline, &ctx,
                                        reinterpret_cast<gsr_stream_t>(stream));
        check(rc, ctx);
    }, py::arg("view"), py::arg("gaussians"), py::arg("num_rendered"), py::arg("radii"), py::arg("geom"), py::arg("binning"),
       py::arg("image"), py::arg("dL_dcolor"), py::arg("dL_dallmap"), py::arg("grads"), py::arg("alloc"), py::arg("stream"));
    m.def("knn3_workspace_bytes", [](int32_t n) { return gsr_knn3_workspace_bytes(n); });
    m.def("knn3", [](uintptr_t xyz, int32_t n, uintptr_t out, uintptr_t ws, size_t ws_bytes, uintptr_t stream) {
        AllocCtx none{py::none(), {}};
        check(gsr_knn3(ptr<const float>(xyz), n, ptr<float>(out), ptr<void>(ws), ws_bytes, reinterpret_cast<gsr_stream_t>(stream)), none);
    });
}
