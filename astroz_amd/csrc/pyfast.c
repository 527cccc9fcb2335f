/* _azfast -- the scalar call of the Python mirror as a CPython function (the reference's Satrec.sgp4 is one too:
 * bindings/python/src/satrec.zig L169-201, 0.4 us per call).  ctypes costs 2-3 us per call before the library is even
 * entered; this module is the same call with the argument handling in C.  It holds no propagation code: `bind` receives
 * the address of libastroz_hip.so's azh_propagate_one_host (from the ctypes handle astroz_amd._native already holds, so
 * whichever library that module loaded is the one called), and `sgp4` calls it for one point.  Optional: without it
 * astroz_amd.api.Satrec.sgp4 makes the same library call through ctypes. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

typedef int32_t (*one_host_fn)(void *c, size_t sat, const double *tsince, size_t n, double *pos, double *vel, uint8_t *err);
static one_host_fn g_one_host = NULL;

static PyObject *az_bind(PyObject *self, PyObject *arg)
{
    (void)self;
    void *p = PyLong_AsVoidPtr(arg);
    if (!p && PyErr_Occurred()) return NULL;
    g_one_host = (one_host_fn)p;
    Py_RETURN_NONE;
}

/* sgp4(handle, sat_index, jd, fr, epoch_jd) -> (tsince_min, rc, err, (x, y, z), (vx, vy, vz));  tsince = ((jd + fr) - epoch) * 1440
 * as satrec.zig L176-178 forms it.  sat_index: the record's row in the handle (Satrec objects made together share one handle) */
static PyObject *az_sgp4(PyObject *self, PyObject *const *args, Py_ssize_t nargs)
{
    (void)self;
    if (nargs != 5) {
        PyErr_SetString(PyExc_TypeError, "sgp4(handle, sat_index, jd, fr, epoch_jd)");
        return NULL;
    }
    if (!g_one_host) {
        PyErr_SetString(PyExc_RuntimeError, "_azfast is not bound to libastroz_hip.so");
        return NULL;
    }
    void *h = PyLong_AsVoidPtr(args[0]);
    const size_t sat = PyLong_AsSize_t(args[1]);
    const double jd = PyFloat_AsDouble(args[2]), fr = PyFloat_AsDouble(args[3]), ep = PyFloat_AsDouble(args[4]);
    if (PyErr_Occurred()) return NULL;
    const double t = ((jd + fr) - ep) * 1440.0;
    double r[3], v[3];
    uint8_t e = 0;
    const int32_t rc = g_one_host(h, sat, &t, 1, r, v, &e);
    return Py_BuildValue("dii(ddd)(ddd)", t, (int)rc, (int)e, r[0], r[1], r[2], v[0], v[1], v[2]);
}

static PyMethodDef methods[] = {
    {"bind", az_bind, METH_O, "bind(address of azh_propagate_one_host)"},
    {"sgp4", (PyCFunction)(void (*)(void))az_sgp4, METH_FASTCALL, "sgp4(handle, sat_index, jd, fr, epoch_jd) -> (tsince, rc, err, r, v)"},
    {NULL, NULL, 0, NULL}};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_azfast", "scalar-call shim over libastroz_hip.so", -1, methods, NULL, NULL, NULL, NULL};
PyMODINIT_FUNC PyInit__azfast(void) { return PyModule_Create(&moddef); }
