/* Host-side packing of the reference's value conventions -- nested tuples of Python ints (curve4q.py) -- into the C ABI's
 * little-endian 64-bit words (include/fourq_amd.h) and back, as a CPython extension: the tuple-level API of fourq_amd.curve4q /
 * fourq_amd.codec spends its time here, not on the GPU (4.5 + 5.6 us per element in Python loops against a
 * kernel that needs 5 ns).  GF(p) values in [0, 2^128) are reduced to [0, p) here, as the reference's `% p1271` (fields.py:29-57) would;
 * anything else (negative, wider) raises OverflowError and the Python caller takes its general path.
 *
 *   pack_fp(points, arity)  -> bytes   points: sequence of `arity`-tuples of (re, im) pairs;    n * arity * 32 bytes
 *   unpack_fp(buffer, arity) -> list   the inverse: list of `arity`-tuples of (re, im) pairs of ints
 *   pack_scalars(seq)       -> bytes   ints in [0, 2^256) -> n * 32 bytes
 *   unpack_scalars(buffer)  -> list
 * Built in-tree by fourq_amd/build.py (gcc, no GPU code); fourq_amd/codec.py falls back to its pure-Python path if it is absent. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>

/* CPython's int internals changed in 3.12 (ob_digit moved, Py_SIZE no longer the sign) and _PyLong_AsByteArray grew an argument in
 * 3.13.  The direct-digit paths below are for 3.10 / 3.11, the interpreters of this image and the only ones these lines have been run
 * on; later versions compile the documented byte-array calls instead (PyLong_AsNativeBytes / PyLong_FromUnsignedNativeBytes from 3.13).
 * If neither compiles, fourq_amd/build.py warns and fourq_amd/codec.py keeps its pure-Python path. */
#if PY_VERSION_HEX < 0x030C0000
#define FQ_DIRECT_DIGITS 1
#else
#define FQ_DIRECT_DIGITS 0
#endif

static int int_to_le(PyObject* v, unsigned char* dst, size_t width) {
    if (!PyLong_Check(v)) {
        PyErr_SetString(PyExc_TypeError, "expected an int");
        return -1;
    }
#if PY_VERSION_HEX >= 0x030D0000
    Py_ssize_t need = PyLong_AsNativeBytes(v, dst, (Py_ssize_t)width,
                                           Py_ASNATIVEBYTES_LITTLE_ENDIAN | Py_ASNATIVEBYTES_UNSIGNED_BUFFER | Py_ASNATIVEBYTES_REJECT_NEGATIVE);
    if (need < 0) {                                        /* negative value: ValueError there, OverflowError for our callers */
        PyErr_Clear();
        PyErr_SetString(PyExc_OverflowError, "negative value");
        return -1;
    }
    if ((size_t)need > width) {
        PyErr_SetString(PyExc_OverflowError, "int too big to convert");
        return -1;
    }
    return 0;
#else
#if FQ_DIRECT_DIGITS
    if (Py_SIZE(v) < 0) {                                  /* negative: the caller reduces */
        PyErr_SetString(PyExc_OverflowError, "negative value");
        return -1;
    }
#endif
    /* unsigned conversion: a negative value raises OverflowError here as well */
    return _PyLong_AsByteArray((PyLongObject*)v, dst, width, 1 /* little endian */, 0 /* unsigned */);
#endif
}

static PyObject* long_from_le(const unsigned char* src, size_t width) {
#if PY_VERSION_HEX >= 0x030D0000
    return PyLong_FromUnsignedNativeBytes(src, width, Py_ASNATIVEBYTES_LITTLE_ENDIAN);
#else
    return _PyLong_FromByteArray(src, width, 1, 0);
#endif
}

/* A GF(p) value, p = 2^127 - 1: any int in [0, 2^128) is stored as its residue in [0, p), as the reference's `% p1271` leaves it. */
static int fp_to_le(PyObject* v, unsigned char* dst) {
    if (int_to_le(v, dst, 16) < 0) return -1;
    uint64_t lo, hi;
    memcpy(&lo, dst, 8);
    memcpy(&hi, dst + 8, 8);
    if (hi >= 0x7fffffffffffffffull) {                     /* possibly >= p: fold bit 127, then one conditional subtraction */
        unsigned __int128 x = ((unsigned __int128)hi << 64) | lo;
        const unsigned __int128 P = (((unsigned __int128)1) << 127) - 1;
        x = (x & P) + (x >> 127);
        if (x >= P) x -= P;
        lo = (uint64_t)x;
        hi = (uint64_t)(x >> 64);
        memcpy(dst, &lo, 8);
        memcpy(dst + 8, &hi, 8);
    }
    return 0;
}

/* 16 little-endian bytes -> int.  With 30-bit digits (every 64-bit CPython) the five digits are written directly. */
static PyObject* long_from_le16(const unsigned char* src) {
#if PYLONG_BITS_IN_DIGIT == 30 && FQ_DIRECT_DIGITS
    uint64_t lo, hi;
    memcpy(&lo, src, 8);
    memcpy(&hi, src + 8, 8);
    if (hi == 0) return PyLong_FromUnsignedLongLong(lo);
    PyLongObject* v = _PyLong_New(5);
    if (!v) return NULL;
    const uint64_t M = (1u << 30) - 1;
    v->ob_digit[0] = (digit)(lo & M);
    v->ob_digit[1] = (digit)((lo >> 30) & M);
    v->ob_digit[2] = (digit)(((lo >> 60) | (hi << 4)) & M);
    v->ob_digit[3] = (digit)((hi >> 26) & M);
    v->ob_digit[4] = (digit)(hi >> 56);
    Py_ssize_t n = 5;
    while (n > 0 && v->ob_digit[n - 1] == 0) n--;
    Py_SET_SIZE(v, n);
    return (PyObject*)v;
#else
    return long_from_le(src, 16);
#endif
}

static PyObject* pack_fp(PyObject* self, PyObject* args) {
    PyObject *points, *fast = NULL, *out = NULL;
    Py_ssize_t arity;
    if (!PyArg_ParseTuple(args, "On", &points, &arity)) return NULL;
    fast = PySequence_Fast(points, "expected a sequence of points");
    if (!fast) return NULL;
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(fast);
    out = PyBytes_FromStringAndSize(NULL, n * arity * 32);
    if (!out) goto fail;
    unsigned char* dst = (unsigned char*)PyBytes_AS_STRING(out);
    for (Py_ssize_t i = 0; i < n; i++) {
        PyObject* P = PySequence_Fast(PySequence_Fast_GET_ITEM(fast, i), "a point is a sequence of GF(p^2) pairs");
        if (!P) goto fail;
        if (PySequence_Fast_GET_SIZE(P) != arity) {
            PyErr_Format(PyExc_ValueError, "expected a point with %zd coordinates, got %zd", arity, PySequence_Fast_GET_SIZE(P));
            Py_DECREF(P);
            goto fail;
        }
        for (Py_ssize_t k = 0; k < arity; k++) {
            PyObject* c = PySequence_Fast(PySequence_Fast_GET_ITEM(P, k), "a GF(p^2) element is a pair");
            if (!c) { Py_DECREF(P); goto fail; }
            if (PySequence_Fast_GET_SIZE(c) != 2) {
                PyErr_SetString(PyExc_ValueError, "a GF(p^2) element is a pair");
                Py_DECREF(c); Py_DECREF(P);
                goto fail;
            }
            if (fp_to_le(PySequence_Fast_GET_ITEM(c, 0), dst) < 0 || fp_to_le(PySequence_Fast_GET_ITEM(c, 1), dst + 16) < 0) {
                Py_DECREF(c); Py_DECREF(P);
                goto fail;
            }
            dst += 32;
            Py_DECREF(c);
        }
        Py_DECREF(P);
    }
    Py_DECREF(fast);
    return out;
fail:
    Py_XDECREF(fast);
    Py_XDECREF(out);
    return NULL;
}

static PyObject* unpack_fp(PyObject* self, PyObject* args) {
    Py_buffer buf;
    Py_ssize_t arity;
    if (!PyArg_ParseTuple(args, "y*n", &buf, &arity)) return NULL;
    PyObject* out = NULL;
    if (arity <= 0 || buf.len % (arity * 32) != 0) {
        PyErr_SetString(PyExc_ValueError, "buffer length is not a multiple of the point size");
        goto done;
    }
    const Py_ssize_t n = buf.len / (arity * 32);
    const unsigned char* src = (const unsigned char*)buf.buf;
    out = PyList_New(n);
    if (!out) goto done;
    for (Py_ssize_t i = 0; i < n; i++) {
        PyObject* P = PyTuple_New(arity);
        if (!P) { Py_CLEAR(out); goto done; }
        PyList_SET_ITEM(out, i, P);
        for (Py_ssize_t k = 0; k < arity; k++, src += 32) {
            PyObject* re = long_from_le16(src);
            PyObject* im = long_from_le16(src + 16);
            PyObject* c = (re && im) ? PyTuple_New(2) : NULL;
            if (!c) { Py_XDECREF(re); Py_XDECREF(im); Py_CLEAR(out); goto done; }
            PyTuple_SET_ITEM(c, 0, re);
            PyTuple_SET_ITEM(c, 1, im);
            PyTuple_SET_ITEM(P, k, c);
        }
    }
done:
    PyBuffer_Release(&buf);
    return out;
}

static PyObject* pack_scalars(PyObject* self, PyObject* arg) {
    PyObject* fast = PySequence_Fast(arg, "expected a sequence of ints");
    if (!fast) return NULL;
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(fast);
    PyObject* out = PyBytes_FromStringAndSize(NULL, n * 32);
    if (out) {
        unsigned char* dst = (unsigned char*)PyBytes_AS_STRING(out);
        for (Py_ssize_t i = 0; i < n; i++, dst += 32)
            if (int_to_le(PySequence_Fast_GET_ITEM(fast, i), dst, 32) < 0) { Py_CLEAR(out); break; }
    }
    Py_DECREF(fast);
    return out;
}

static PyObject* unpack_scalars(PyObject* self, PyObject* args) {
    Py_buffer buf;
    if (!PyArg_ParseTuple(args, "y*", &buf)) return NULL;
    PyObject* out = NULL;
    if (buf.len % 32 != 0) {
        PyErr_SetString(PyExc_ValueError, "buffer length is not a multiple of 32");
    } else {
        const Py_ssize_t n = buf.len / 32;
        out = PyList_New(n);
        for (Py_ssize_t i = 0; out && i < n; i++) {
            PyObject* v = long_from_le((const unsigned char*)buf.buf + 32 * i, 32);
            if (!v) { Py_CLEAR(out); break; }
            PyList_SET_ITEM(out, i, v);
        }
    }
    PyBuffer_Release(&buf);
    return out;
}

static PyMethodDef methods[] = {
    { "pack_fp", pack_fp, METH_VARARGS, "sequence of arity-tuples of (re, im) pairs -> bytes (32 per GF(p^2) element)" },
    { "unpack_fp", unpack_fp, METH_VARARGS, "buffer, arity -> list of arity-tuples of (re, im) pairs" },
    { "pack_scalars", pack_scalars, METH_O, "ints in [0, 2^256) -> bytes (32 each, little endian)" },
    { "unpack_scalars", unpack_scalars, METH_VARARGS, "buffer -> list of ints" },
    { NULL, NULL, 0, NULL }
};
static struct PyModuleDef module = { PyModuleDef_HEAD_INIT, "_fastcodec", "tuple <-> word packing for fourq_amd.codec", -1, methods };
PyMODINIT_FUNC PyInit__fastcodec(void) { return PyModule_Create(&module); }
