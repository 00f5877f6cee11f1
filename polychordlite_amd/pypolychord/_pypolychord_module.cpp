// _pypolychord -- compiled CPython extension over libpolychord_hip.so.
//
// Keeps the one-function surface of the reference's extension module (reference
// pypolychord/_pypolychord.cpp:119-228): `run(*37 positional)` with the parse format
// "OOOiiiiiiiiddidiiiiiiiiiiidissO!O!O!i", the same TypeError / ValueError checks (:178-204),
// callbacks handed numpy views of engine-owned buffers (theta and cube read-only, :29-115), `None` returned.
// The work is done by `polychord_c_interface` of the HIP engine (include/polychord_hip.h part 1), which the
// module binds at load time: no Fortran, no MPI, no CPU sampler behind it.
//
// Differences that are deliberate:
//  * a Python exception in a callback does not unwind through the engine (the reference throws a C++
//    exception through the Fortran frames, :219-224): the error is parked, the engine is asked to stop at its
//    next host boundary (polychord_hip_request_stop) and `run` re-raises it;
//  * a callable carrying a `symbol` attribute (polychordlite_amd.pypolychord.device_likelihoods) is resolved to
//    the library's own function of that name, so the likelihood runs fused inside the slice kernel.
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#define NPY_NO_DEPRECATED_API NPY_1_7_API_VERSION
#include <numpy/arrayobject.h>
#include <dlfcn.h>
#include <algorithm>
#include <string>
#include <vector>
#include "polychord_hip.h"

namespace {

struct Bridge {                 // one run at a time per process, like the reference's module statics (:27,61,83)
    PyObject *like = nullptr, *prior = nullptr, *dumper = nullptr;
    PyObject *err_type = nullptr, *err_value = nullptr, *err_tb = nullptr;
    double logzero = -1e30;
    bool failed() const { return err_type != nullptr; }
    void park_error()
    {   // keep the first error, ask the engine to wind down
        if (!failed()) PyErr_Fetch(&err_type, &err_value, &err_tb); else PyErr_Clear();
        polychord_hip_request_stop();
    }
};
Bridge g_bridge;

PyObject *view1(double *p, int n, bool writeable)
{
    npy_intp shape[1] = { n };
    PyObject *a = PyArray_SimpleNewFromData(1, shape, NPY_DOUBLE, (void *)p);
    if (a && !writeable) PyArray_CLEARFLAGS(reinterpret_cast<PyArrayObject *>(a), NPY_ARRAY_WRITEABLE);
    return a;
}
PyObject *view2(double *p, int rows, int cols)
{
    npy_intp shape[2] = { rows, cols };
    PyObject *a = PyArray_SimpleNewFromData(2, shape, NPY_DOUBLE, (void *)p);
    if (a) PyArray_CLEARFLAGS(reinterpret_cast<PyArrayObject *>(a), NPY_ARRAY_WRITEABLE);
    return a;
}

// loglikelihood(theta, phi) -> float; phi filled in place (_pypolychord.cpp:29-58)
double cb_loglike(double *theta, int nDims, double *phi, int nDerived)
{
    Bridge &b = g_bridge;
    if (b.failed()) return b.logzero;
    PyObject *t = view1(theta, nDims, false), *p = t ? view1(phi, nDerived, true) : nullptr;
    if (!t || !p) { Py_XDECREF(t); Py_XDECREF(p); b.park_error(); return b.logzero; }
    PyObject *r = PyObject_CallFunctionObjArgs(b.like, t, p, nullptr);
    Py_DECREF(t); Py_DECREF(p);
    if (!r) { b.park_error(); return b.logzero; }
    if (!PyFloat_Check(r)) {
        Py_DECREF(r);
        PyErr_SetString(PyExc_TypeError, "loglikelihood must be a float (element 0 of loglikelihood return)");
        b.park_error();
        return b.logzero;
    }
    const double v = PyFloat_AsDouble(r);
    Py_DECREF(r);
    return v;
}

// prior(cube, theta): theta written in place (_pypolychord.cpp:63-80)
void cb_prior(double *cube, double *theta, int nDims)
{
    Bridge &b = g_bridge;
    if (b.failed()) return;
    PyObject *c = view1(cube, nDims, false), *t = c ? view1(theta, nDims, true) : nullptr;
    if (!c || !t) { Py_XDECREF(c); Py_XDECREF(t); b.park_error(); return; }
    PyObject *r = PyObject_CallFunctionObjArgs(b.prior, c, t, nullptr);
    Py_DECREF(c); Py_DECREF(t);
    if (!r) b.park_error(); else Py_DECREF(r);
}

// dumper(live, dead, logweights, logZ, logZerr) (_pypolychord.cpp:85-115)
void cb_dumper(int ndead, int nlive, int npars, double *live, double *dead, double *logweights, double logZ, double logZerr)
{
    Bridge &b = g_bridge;
    if (b.failed()) return;
    PyObject *l = view2(live, nlive, npars), *d = l ? view2(dead, ndead, npars) : nullptr;
    PyObject *w = d ? view1(logweights, ndead, false) : nullptr;
    PyObject *z = w ? PyFloat_FromDouble(logZ) : nullptr, *e = z ? PyFloat_FromDouble(logZerr) : nullptr;
    if (e) {
        PyObject *r = PyObject_CallFunctionObjArgs(b.dumper, l, d, w, z, e, nullptr);
        if (!r) b.park_error(); else Py_DECREF(r);
    } else b.park_error();
    Py_XDECREF(l); Py_XDECREF(d); Py_XDECREF(w); Py_XDECREF(z); Py_XDECREF(e);
}

// handle of the engine library this module is linked against (Python loads extensions RTLD_LOCAL, so the default
// search scope does not see it)
void *engine_handle()
{
    static void *h = nullptr;
    if (!h) {
        Dl_info info;
        if (dladdr((void *)&polychord_c_interface, &info) && info.dli_fname) h = dlopen(info.dli_fname, RTLD_NOW | RTLD_NOLOAD);
    }
    return h;
}

// A device functor: `obj.symbol` names an exported function of the engine library, `obj.configure(nDims)` sets its
// process-global parameters.  Returns nullptr (no error set) for ordinary callables.
void *device_symbol(PyObject *callable, int nDims, bool &error)
{
    error = false;
    PyObject *obj = PyObject_GetAttrString(callable, "__wrapped_builtin__");
    if (!obj) { PyErr_Clear(); obj = callable; Py_INCREF(obj); }
    PyObject *sym = PyObject_GetAttrString(obj, "symbol");
    if (!sym) { PyErr_Clear(); Py_DECREF(obj); return nullptr; }
    void *fn = nullptr;
    if (PyUnicode_Check(sym)) {
        PyObject *r = PyObject_CallMethod(obj, "configure", "i", nDims);
        if (!r) error = true; else {
            Py_DECREF(r);
            const char *name = PyUnicode_AsUTF8(sym);
            fn = (name && engine_handle()) ? dlsym(engine_handle(), name) : nullptr;
            if (!fn) { PyErr_Format(PyExc_AttributeError, "libpolychord_hip.so exports no function '%s'", name ? name : "?"); error = true; }
        }
    }
    Py_DECREF(sym); Py_DECREF(obj);
    return fn;
}

PyObject *py_run(PyObject *, PyObject *args)
{
    PyObject *f_like, *f_prior, *f_dumper, *l_frac, *l_dims, *d_nlives;
    int nDims, nDerived, nlive, num_repeats, nprior, nfail, do_clustering, feedback, max_ndead;
    int posteriors, equals, cluster_posteriors, write_resume, write_paramnames, read_resume, write_stats, write_live,
        write_dead, write_prior, maximise, synchronous, seed;
    double precision_criterion, logzero, boost_posterior, compression_factor;
    const char *base_dir, *file_root;
    if (!PyArg_ParseTuple(args, "OOOiiiiiiiiddidiiiiiiiiiiidissO!O!O!i:run", &f_like, &f_prior, &f_dumper, &nDims, &nDerived,
                          &nlive, &num_repeats, &nprior, &nfail, &do_clustering, &feedback, &precision_criterion, &logzero,
                          &max_ndead, &boost_posterior, &posteriors, &equals, &cluster_posteriors, &write_resume,
                          &write_paramnames, &read_resume, &write_stats, &write_live, &write_dead, &write_prior, &maximise,
                          &compression_factor, &synchronous, &base_dir, &file_root, &PyList_Type, &l_frac, &PyList_Type,
                          &l_dims, &PyDict_Type, &d_nlives, &seed))
        return nullptr;
    for (PyObject *f : { f_like, f_prior, f_dumper })
        if (!PyCallable_Check(f)) { PyErr_SetString(PyExc_TypeError, "loglikelihood, prior and dumper must be callable"); return nullptr; }
    // grades: two lists of one length, dims summing to nDims (_pypolychord.cpp:178-204)
    std::vector<double> grade_frac; std::vector<int> grade_dims;
    for (Py_ssize_t i = 0; i < PyList_Size(l_frac); ++i) {
        const double v = PyFloat_AsDouble(PyList_GetItem(l_frac, i));
        if (v == -1.0 && PyErr_Occurred()) { PyErr_SetString(PyExc_TypeError, "grade_frac must be a list of doubles"); return nullptr; }
        grade_frac.push_back(v);
    }
    for (Py_ssize_t i = 0; i < PyList_Size(l_dims); ++i) {
        PyObject *it = PyList_GetItem(l_dims, i);
        const long v = PyLong_Check(it) ? PyLong_AsLong(it) : -1;
        if (!PyLong_Check(it) || (v == -1 && PyErr_Occurred())) { PyErr_Clear(); PyErr_SetString(PyExc_TypeError, "grade_dims must be a list of integers"); return nullptr; }
        grade_dims.push_back((int)v);
    }
    if (grade_frac.size() != grade_dims.size()) { PyErr_SetString(PyExc_ValueError, "grade_dims and grade_frac must have the same size"); return nullptr; }
    long tot = 0;
    for (int v : grade_dims) tot += v;
    if (tot != nDims) { PyErr_SetString(PyExc_ValueError, "grade_dims must sum to nDims"); return nullptr; }
    // nlives: {logL contour: number of live points}, handed over in increasing contour order
    std::vector<std::pair<double, int>> dyn;
    {
        PyObject *k, *v; Py_ssize_t pos = 0;
        while (PyDict_Next(d_nlives, &pos, &k, &v)) {
            const double kk = PyFloat_AsDouble(k);
            const long vv = PyLong_AsLong(v);
            if ((kk == -1.0 || vv == -1) && PyErr_Occurred()) { PyErr_Clear(); PyErr_SetString(PyExc_TypeError, "nlives must be a dict mapping floats to integers"); return nullptr; }
            dyn.push_back({ kk, (int)vv });
        }
        std::sort(dyn.begin(), dyn.end());
    }
    std::vector<double> loglikes; std::vector<int> nlives;
    for (auto &p : dyn) { loglikes.push_back(p.first); nlives.push_back(p.second); }

    bool err = false;
    void *dev_like = device_symbol(f_like, nDims, err);
    if (err) return nullptr;
    void *dev_prior = device_symbol(f_prior, nDims, err);
    if (err) return nullptr;

    Bridge &b = g_bridge;
    Py_XDECREF(b.like); Py_XDECREF(b.prior); Py_XDECREF(b.dumper);
    b = Bridge{};
    b.like = f_like; b.prior = f_prior; b.dumper = f_dumper; b.logzero = logzero;
    Py_INCREF(f_like); Py_INCREF(f_prior); Py_INCREF(f_dumper);

    std::string base = base_dir, root = file_root;
    int comm = 0;
    polychord_hip_set_option("halt_returns", 1.0);
    // the GIL stays with this thread for the whole run: callbacks are made synchronously from it (_pypolychord.cpp:219)
    polychord_c_interface(dev_like ? (polychord_loglike_fn)dev_like : cb_loglike, dev_prior ? (polychord_prior_fn)dev_prior : cb_prior,
                          cb_dumper, nlive, num_repeats, nprior, nfail, do_clustering != 0, feedback, precision_criterion, logzero,
                          max_ndead, boost_posterior, posteriors != 0, equals != 0, cluster_posteriors != 0, write_resume != 0,
                          write_paramnames != 0, read_resume != 0, write_stats != 0, write_live != 0, write_dead != 0,
                          write_prior != 0, maximise != 0, compression_factor, synchronous != 0, nDims, nDerived,
                          (char *)base.c_str(), (char *)root.c_str(), (int)grade_frac.size(), grade_frac.data(), grade_dims.data(),
                          (int)loglikes.size(), loglikes.empty() ? nullptr : loglikes.data(), nlives.empty() ? nullptr : nlives.data(),
                          seed, &comm);
    if (b.failed()) {
        PyErr_Restore(b.err_type, b.err_value, b.err_tb);
        b.err_type = b.err_value = b.err_tb = nullptr;
        return nullptr;
    }
    // a fatal condition of the engine (the reference prints and stops the process, abort.F90:19-29; inside an
    // interpreter that is an exception)
    if (const char *msg = polychord_hip_last_error()) { PyErr_SetString(PyExc_RuntimeError, msg); return nullptr; }
    Py_RETURN_NONE;
}

PyMethodDef methods[] = {
    { "run", py_run, METH_VARARGS,
      "run(loglikelihood, prior, dumper, nDims, nDerived, nlive, num_repeats, nprior, nfail, do_clustering, feedback, "
      "precision_criterion, logzero, max_ndead, boost_posterior, posteriors, equals, cluster_posteriors, write_resume, "
      "write_paramnames, read_resume, write_stats, write_live, write_dead, write_prior, maximise, compression_factor, "
      "synchronous, base_dir, file_root, grade_frac, grade_dims, nlives, seed) -> None" },
    { nullptr, nullptr, 0, nullptr }
};

PyModuleDef moduledef = { PyModuleDef_HEAD_INIT, "_pypolychord",
                          "PolyChordLite's _pypolychord entry point on the MI355X HIP engine (libpolychord_hip.so)", -1, methods,
                          nullptr, nullptr, nullptr, nullptr };

}  // namespace

PyMODINIT_FUNC PyInit__pypolychord(void)
{
    import_array();
    PyObject *m = PyModule_Create(&moduledef);
    if (m) PyModule_AddStringConstant(m, "backend", "libpolychord_hip.so (HIP, gfx950)");
    return m;
}
