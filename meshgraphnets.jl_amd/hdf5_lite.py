"""A small binding of libhdf5 (the C library HDF5.jl wraps; reference `import HDF5: ...`, src/dataset.jl:6-10, and `h5open` /
`create_group` / `open_group` / `keys` / `read` as used in src/dataset.jl:118-352 and src/MeshGraphNets.jl:638-669).

Only what the reference's dataset branches and its evaluation writer need: open / create a file, list a group's links in name
order (what `keys(::HDF5.Group)` returns), read a dataset of integer / float / compound / fixed-array type into numpy, write
integer / float datasets, create groups.  JLD2 files are HDF5 files (JLD2.jl writes plain numeric arrays as ordinary datasets), so the
`.jld2` branch goes through the same calls.

Array convention: HDF5 stores dimensions slowest-first (C order); HDF5.jl and JLD2.jl present the SAME bytes to Julia with the
dimensions reversed (column-major).  `read` returns the C-order numpy array; `.T` of it is the Julia view.

The library is bound at run time (ctypes): `MGN_HDF5_LIB`, then the loader's search path, then the locations this image and the common
distributions use.  Nothing here falls back to another format: without the library every entry point raises `Hdf5Unavailable`."""
import ctypes as C
import ctypes.util
import os

import numpy as np

hid_t = C.c_int64          # HDF5 >= 1.10
hsize_t = C.c_uint64
herr_t = C.c_int

H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5P_DEFAULT, H5S_ALL = 0, 0
H5_INDEX_NAME, H5_ITER_INC = 0, 0
H5T_INTEGER, H5T_FLOAT, H5T_STRING, H5T_COMPOUND, H5T_ENUM, H5T_ARRAY = 0, 1, 3, 6, 8, 10
H5T_DIR_ASCEND = 1
H5T_SGN_NONE = 0


class Hdf5Unavailable(RuntimeError):
    pass


class Hdf5Error(RuntimeError):
    pass


class _GInfo(C.Structure):   # H5G_info_t
    _fields_ = [("storage_type", C.c_int), ("nlinks", hsize_t), ("max_corder", C.c_int64), ("mounted", C.c_uint)]


_CANDIDATES = (
    "/opt/conda/lib/libhdf5.so", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so", "/usr/lib/x86_64-linux-gnu/libhdf5_serial.so",
    "/usr/lib64/libhdf5.so", "/usr/local/lib/libhdf5.so",
)
_lib = None


def library_path():
    """Where libhdf5 would be loaded from, or None."""
    env = os.environ.get("MGN_HDF5_LIB")
    if env:
        return env
    found = ctypes.util.find_library("hdf5") or ctypes.util.find_library("hdf5_serial")
    if found:
        return found
    for p in _CANDIDATES:
        if os.path.exists(p):
            return p
    return None


def available():
    try:
        _load()
        return True
    except Hdf5Unavailable:
        return False


def _load():
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if path is None:
        raise Hdf5Unavailable("libhdf5 not found (set MGN_HDF5_LIB): the .h5 / .jld2 dataset branches and trajectories.h5 need it")
    try:
        lib = C.CDLL(path)
    except OSError as e:
        raise Hdf5Unavailable(f"cannot load {path}: {e}") from e

    def sig(name, res, *args):
        f = getattr(lib, name)
        f.restype, f.argtypes = res, list(args)
        return f

    sig("H5open", herr_t)
    sig("H5get_libversion", herr_t, C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_uint))
    sig("H5Eset_auto2", herr_t, hid_t, C.c_void_p, C.c_void_p)
    sig("H5Fopen", hid_t, C.c_char_p, C.c_uint, hid_t)
    sig("H5Fcreate", hid_t, C.c_char_p, C.c_uint, hid_t, hid_t)
    sig("H5Fclose", herr_t, hid_t)
    sig("H5Fflush", herr_t, hid_t, C.c_int)
    sig("H5Gopen2", hid_t, hid_t, C.c_char_p, hid_t)
    sig("H5Gcreate2", hid_t, hid_t, C.c_char_p, hid_t, hid_t, hid_t)
    sig("H5Gclose", herr_t, hid_t)
    sig("H5Gget_info", herr_t, hid_t, C.POINTER(_GInfo))
    sig("H5Lget_name_by_idx", C.c_ssize_t, hid_t, C.c_char_p, C.c_int, C.c_int, hsize_t, C.c_char_p, C.c_size_t, hid_t)
    sig("H5Lexists", C.c_int, hid_t, C.c_char_p, hid_t)
    sig("H5Dopen2", hid_t, hid_t, C.c_char_p, hid_t)
    sig("H5Dcreate2", hid_t, hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t)
    sig("H5Dclose", herr_t, hid_t)
    sig("H5Dget_space", hid_t, hid_t)
    sig("H5Dget_type", hid_t, hid_t)
    sig("H5Dread", herr_t, hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p)
    sig("H5Dwrite", herr_t, hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p)
    sig("H5Screate_simple", hid_t, C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t))
    sig("H5Sget_simple_extent_ndims", C.c_int, hid_t)
    sig("H5Sget_simple_extent_dims", C.c_int, hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t))
    sig("H5Sclose", herr_t, hid_t)
    sig("H5Tclose", herr_t, hid_t)
    sig("H5Tget_class", C.c_int, hid_t)
    sig("H5Tget_size", C.c_size_t, hid_t)
    sig("H5Tget_sign", C.c_int, hid_t)
    sig("H5Tget_native_type", hid_t, hid_t, C.c_int)
    sig("H5Tget_nmembers", C.c_int, hid_t)
    sig("H5Tget_member_name", C.c_void_p, hid_t, C.c_uint)
    sig("H5Tget_member_offset", C.c_size_t, hid_t, C.c_uint)
    sig("H5Tget_member_type", hid_t, hid_t, C.c_uint)
    sig("H5Tget_super", hid_t, hid_t)
    sig("H5Tget_array_ndims", C.c_int, hid_t)
    sig("H5Tget_array_dims2", C.c_int, hid_t, C.POINTER(hsize_t))
    sig("H5free_memory", herr_t, C.c_void_p)

    if lib.H5open() < 0:
        raise Hdf5Unavailable(f"H5open failed in {path}")
    maj, mnr, rel = C.c_uint(), C.c_uint(), C.c_uint()
    lib.H5get_libversion(C.byref(maj), C.byref(mnr), C.byref(rel))
    if (maj.value, mnr.value) < (1, 10):
        raise Hdf5Unavailable(f"{path} is HDF5 {maj.value}.{mnr.value}.{rel.value}; 1.10 or later is needed (64-bit identifiers)")
    lib.H5Eset_auto2(0, None, None)       # failures are reported through return codes -> exceptions, not printed stacks
    lib._version = (maj.value, mnr.value, rel.value)
    lib._native = {
        np.dtype(k): hid_t.in_dll(lib, f"H5T_NATIVE_{v}_g").value
        for k, v in {"int8": "INT8", "uint8": "UINT8", "int16": "INT16", "uint16": "UINT16", "int32": "INT32", "uint32": "UINT32",
                     "int64": "INT64", "uint64": "UINT64", "float32": "FLOAT", "float64": "DOUBLE"}.items()
    }
    _lib = lib
    return lib


def version():
    return _load()._version


def _np_dtype(lib, t):
    """numpy dtype of a NATIVE HDF5 datatype (integer, float, enum over integers, compound, fixed-size array)."""
    cls, size = lib.H5Tget_class(t), lib.H5Tget_size(t)
    if cls == H5T_INTEGER:
        return np.dtype(("u" if lib.H5Tget_sign(t) == H5T_SGN_NONE else "i") + str(size))
    if cls == H5T_FLOAT:
        if size not in (2, 4, 8):
            raise Hdf5Error(f"{size}-byte float type")
        return np.dtype("f" + str(size))
    if cls == H5T_ENUM:              # (h5py writes numpy bool as an enum over int8)
        sup = lib.H5Tget_super(t)
        try:
            return _np_dtype(lib, sup)
        finally:
            lib.H5Tclose(sup)
    if cls == H5T_COMPOUND:
        names, fmts, offs = [], [], []
        for i in range(lib.H5Tget_nmembers(t)):
            p = lib.H5Tget_member_name(t, i)
            names.append(C.string_at(p).decode())
            lib.H5free_memory(p)
            mt = lib.H5Tget_member_type(t, i)
            try:
                fmts.append(_np_dtype(lib, mt))
            finally:
                lib.H5Tclose(mt)
            offs.append(lib.H5Tget_member_offset(t, i))
        return np.dtype({"names": names, "formats": fmts, "offsets": offs, "itemsize": size})
    if cls == H5T_ARRAY:
        nd = lib.H5Tget_array_ndims(t)
        dims = (hsize_t * nd)()
        lib.H5Tget_array_dims2(t, dims)
        sup = lib.H5Tget_super(t)
        try:
            return np.dtype((_np_dtype(lib, sup), tuple(int(d) for d in dims)))
        finally:
            lib.H5Tclose(sup)
    raise Hdf5Error(f"unsupported HDF5 datatype class {cls}")


class Group:
    """`HDF5.Group` / `HDF5.File` as the reference uses them: `keys`, `haskey`, `open_group` (`g[name]`), `read`, `create_group`,
    `g[name] = array`."""

    def __init__(self, lib, gid, name, owner=None):
        self._lib, self._id, self.name, self._owner = lib, gid, name, owner

    # ---- reading ----
    def keys(self):
        info = _GInfo()
        if self._lib.H5Gget_info(self._id, C.byref(info)) < 0:
            raise Hdf5Error(f"H5Gget_info({self.name})")
        out = []
        for i in range(info.nlinks):
            n = self._lib.H5Lget_name_by_idx(self._id, b".", H5_INDEX_NAME, H5_ITER_INC, i, None, 0, H5P_DEFAULT)
            if n < 0:
                raise Hdf5Error(f"H5Lget_name_by_idx({self.name}, {i})")
            buf = C.create_string_buffer(n + 1)
            self._lib.H5Lget_name_by_idx(self._id, b".", H5_INDEX_NAME, H5_ITER_INC, i, buf, n + 1, H5P_DEFAULT)
            out.append(buf.value.decode())
        return out

    def __len__(self):
        return len(self.keys())

    def __contains__(self, name):
        return self._lib.H5Lexists(self._id, name.encode(), H5P_DEFAULT) > 0

    def open_group(self, name):
        gid = self._lib.H5Gopen2(self._id, name.encode(), H5P_DEFAULT)
        if gid < 0:
            raise KeyError(f"group '{name}' not found in '{self.name}'")
        return Group(self._lib, gid, self.name.rstrip("/") + "/" + name, owner=self)

    def read(self, name):
        """`read(group, name)`: the whole dataset as a C-order numpy array (shape = the HDF5 dimensions; a scalar dataspace gives a
        0-d array).  The file's type is converted to the matching native type by the library."""
        lib = self._lib
        did = lib.H5Dopen2(self._id, name.encode(), H5P_DEFAULT)
        if did < 0:
            raise KeyError(f"dataset '{name}' not found in '{self.name}'")
        sid = tid = nat = -1
        try:
            sid = lib.H5Dget_space(did)
            nd = lib.H5Sget_simple_extent_ndims(sid)
            if nd < 0:
                raise Hdf5Error(f"dataspace of '{name}'")
            dims = (hsize_t * max(nd, 1))()
            if nd:
                lib.H5Sget_simple_extent_dims(sid, dims, None)
            shape = tuple(int(dims[i]) for i in range(nd))
            tid = lib.H5Dget_type(did)
            nat = lib.H5Tget_native_type(tid, H5T_DIR_ASCEND)
            if nat < 0:
                raise Hdf5Error(f"datatype of '{name}' has no native equivalent")
            out = np.empty(shape, dtype=_np_dtype(lib, nat))
            if out.size and lib.H5Dread(did, nat, H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data_as(C.c_void_p)) < 0:
                raise Hdf5Error(f"H5Dread('{name}')")
            return out
        finally:
            for closer, ident in ((lib.H5Tclose, nat), (lib.H5Tclose, tid), (lib.H5Sclose, sid), (lib.H5Dclose, did)):
                if ident >= 0:
                    closer(ident)

    def rank(self, name):
        """number of dimensions of the dataset's DATASPACE (a vector of array-typed elements has rank 1 and reads as a 2-d numpy array)"""
        lib = self._lib
        did = lib.H5Dopen2(self._id, name.encode(), H5P_DEFAULT)
        if did < 0:
            raise KeyError(f"dataset '{name}' not found in '{self.name}'")
        sid = lib.H5Dget_space(did)
        try:
            return int(lib.H5Sget_simple_extent_ndims(sid))
        finally:
            lib.H5Sclose(sid)
            lib.H5Dclose(did)

    def __getitem__(self, name):
        """`file[k]`: a group if the link is one, else the dataset's contents."""
        gid = self._lib.H5Gopen2(self._id, name.encode(), H5P_DEFAULT)
        if gid >= 0:
            return Group(self._lib, gid, self.name.rstrip("/") + "/" + name, owner=self)
        return self.read(name)

    # ---- writing ----
    def create_group(self, name):
        gid = self._lib.H5Gcreate2(self._id, name.encode(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT)
        if gid < 0:
            raise Hdf5Error(f"cannot create group '{name}' in '{self.name}'")
        return Group(self._lib, gid, self.name.rstrip("/") + "/" + name, owner=self)

    def write(self, name, array):
        """`group[name] = array`: a contiguous dataset of the array's dtype with the array's C-order shape."""
        lib = self._lib
        a = np.ascontiguousarray(array)
        if a.dtype == np.bool_:
            a = a.astype(np.uint8)
        if a.dtype not in lib._native:
            raise Hdf5Error(f"cannot write dtype {a.dtype}")
        t = lib._native[a.dtype]
        dims = (hsize_t * max(a.ndim, 1))(*a.shape)
        sid = lib.H5Screate_simple(a.ndim, dims, None)
        if sid < 0:
            raise Hdf5Error("H5Screate_simple")
        did = lib.H5Dcreate2(self._id, name.encode(), t, sid, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT)
        try:
            if did < 0:
                raise Hdf5Error(f"cannot create dataset '{name}' in '{self.name}'")
            if a.size and lib.H5Dwrite(did, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, a.ctypes.data_as(C.c_void_p)) < 0:
                raise Hdf5Error(f"H5Dwrite('{name}')")
        finally:
            if did >= 0:
                lib.H5Dclose(did)
            lib.H5Sclose(sid)

    __setitem__ = write

    def close(self):
        if self._id >= 0:
            self._lib.H5Gclose(self._id)
            self._id = -1

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class File(Group):
    """`h5open(path, "r" | "w")` (also `jldopen(path, "r")` for the numeric arrays the dataset branch reads)."""

    def __init__(self, path, mode="r"):
        lib = _load()
        p = os.fsencode(str(path))
        if mode == "r":
            if not os.path.isfile(path):
                raise FileNotFoundError(path)
            fid = lib.H5Fopen(p, H5F_ACC_RDONLY, H5P_DEFAULT)
        elif mode == "w":
            fid = lib.H5Fcreate(p, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
        else:
            raise ValueError("mode must be 'r' or 'w'")
        if fid < 0:
            raise Hdf5Error(f"cannot open {path} (mode {mode}): not an HDF5 file, or not accessible")
        super().__init__(lib, fid, "/")
        self.path = str(path)

    def close(self):
        if self._id >= 0:
            self._lib.H5Fclose(self._id)
            self._id = -1
