"""Non-executing reader for motion/terrain pickles.

The PARC motion format (reference: zmotion_editing_tools/motion_edit_lib.py:189-225,
README.md:90-111) is a pickle dict ``{fps, loop_mode, frames, contacts, terrain: SubTerrain}``
whose leaves are numpy arrays.  ``pickle.load`` executes whatever the file names; this reader
never does.  It walks the opcode stream with ``pickletools.genops`` (a pure disassembler), keeps an
inert stack of records (``Global``, ``Call``, ``Obj``) and afterwards materialises ONLY numpy arrays /
dtypes / scalars from the raw bytes it finds, and torch tensors through ``torch.load(weights_only=True)``
of their storage blob.  Anything else stays an inert record.

Use it for files you did not write (e.g. the clips that ship with the reference); files this
package wrote itself can go through ``pickle`` as the reference does.
"""
import pickletools
from collections import OrderedDict

import numpy as np


class Global:
    __slots__ = ("module", "name")

    def __init__(self, module, name):
        self.module = module
        self.name = name

    def __repr__(self):
        return "Global({}.{})".format(self.module, self.name)


class Call:
    """A REDUCE that was *not* executed: callable record + args (+ BUILD state)."""
    __slots__ = ("func", "args", "state")

    def __init__(self, func, args):
        self.func = func
        self.args = args
        self.state = None


class Obj:
    """A NEWOBJ that was *not* executed: class record + args (+ BUILD state)."""
    __slots__ = ("cls", "args", "state")

    def __init__(self, cls, args):
        self.cls = cls
        self.args = args
        self.state = None


class _Mark:
    pass


_MARK = _Mark()


def _pop_to_mark(stack):
    items = []
    while True:
        x = stack.pop()
        if x is _MARK:
            break
        items.append(x)
    items.reverse()
    return items


def load_inert(data):
    """Return the object tree of a pickle byte string without executing anything."""
    stack = []
    memo = {}
    for op, arg, _pos in pickletools.genops(data):
        n = op.name
        if n in ("PROTO", "FRAME"):
            continue
        elif n == "STOP":
            break
        elif n == "MARK":
            stack.append(_MARK)
        elif n in ("MEMOIZE",):
            memo[len(memo)] = stack[-1]
        elif n in ("BINPUT", "LONG_BINPUT", "PUT"):
            memo[int(arg)] = stack[-1]
        elif n in ("BINGET", "LONG_BINGET", "GET"):
            stack.append(memo[int(arg)])
        elif n in ("EMPTY_DICT",):
            stack.append(OrderedDict())
        elif n in ("EMPTY_LIST",):
            stack.append([])
        elif n in ("EMPTY_TUPLE",):
            stack.append(())
        elif n in ("NONE",):
            stack.append(None)
        elif n == "NEWTRUE":
            stack.append(True)
        elif n == "NEWFALSE":
            stack.append(False)
        elif n in ("BININT", "BININT1", "BININT2", "LONG1", "LONG4", "INT", "LONG", "BINFLOAT", "FLOAT",
                   "SHORT_BINUNICODE", "BINUNICODE", "BINUNICODE8", "UNICODE",
                   "SHORT_BINBYTES", "BINBYTES", "BINBYTES8", "BYTEARRAY8",
                   "SHORT_BINSTRING", "BINSTRING", "STRING"):
            stack.append(arg)
        elif n == "TUPLE1":
            a = stack.pop()
            stack.append((a,))
        elif n == "TUPLE2":
            b = stack.pop()
            a = stack.pop()
            stack.append((a, b))
        elif n == "TUPLE3":
            c = stack.pop()
            b = stack.pop()
            a = stack.pop()
            stack.append((a, b, c))
        elif n == "TUPLE":
            stack.append(tuple(_pop_to_mark(stack)))
        elif n == "LIST":
            stack.append(list(_pop_to_mark(stack)))
        elif n == "DICT":
            items = _pop_to_mark(stack)
            stack.append(OrderedDict(zip(items[0::2], items[1::2])))
        elif n == "APPEND":
            v = stack.pop()
            stack[-1].append(v)
        elif n == "APPENDS":
            items = _pop_to_mark(stack)
            stack[-1].extend(items)
        elif n == "SETITEM":
            v = stack.pop()
            k = stack.pop()
            stack[-1][k] = v
        elif n == "SETITEMS":
            items = _pop_to_mark(stack)
            d = stack[-1]
            for k, v in zip(items[0::2], items[1::2]):
                d[k] = v
        elif n == "STACK_GLOBAL":
            name = stack.pop()
            module = stack.pop()
            stack.append(Global(module, name))
        elif n == "GLOBAL":
            module, name = arg.split(" ")
            stack.append(Global(module, name))
        elif n == "REDUCE":
            args = stack.pop()
            func = stack.pop()
            if isinstance(func, Global) and func.module == "collections" and func.name == "OrderedDict" and args == ():
                stack.append(OrderedDict())      # an empty ordered dict is data, not code
            else:
                stack.append(Call(func, args))
        elif n in ("NEWOBJ",):
            args = stack.pop()
            cls = stack.pop()
            stack.append(Obj(cls, args))
        elif n == "BUILD":
            state = stack.pop()
            tgt = stack[-1]
            if isinstance(tgt, (Call, Obj)):
                tgt.state = state
            else:
                raise ValueError("BUILD on unsupported target {!r}".format(type(tgt)))
        else:
            raise ValueError("unsupported pickle opcode {} (file is not a plain motion pickle)".format(n))
    assert len(stack) == 1, "malformed pickle"
    return stack[0]


def _is_global(x, module_suffixes, name):
    return isinstance(x, Global) and x.name == name and any(x.module == m or x.module.endswith(m) for m in module_suffixes)


def _dtype_from(rec):
    # Call(Global numpy.dtype, ('f4', False, True)) with state (3, '<', None, None, None, -1, -1, 0)
    if not (isinstance(rec, Call) and _is_global(rec.func, ("numpy",), "dtype")):
        raise ValueError("unsupported dtype record")
    code = rec.args[0]
    endian = "<"
    if rec.state is not None and isinstance(rec.state, tuple) and len(rec.state) > 1 and rec.state[1] in ("<", ">", "|", "="):
        endian = rec.state[1]
    if endian in ("|", "="):
        return np.dtype(code)
    return np.dtype(endian + code)


class Unresolved:
    """Placeholder for a record this reader refuses to evaluate (e.g. a torch tensor rebuild)."""

    def __init__(self, what):
        self.what = what

    def __repr__(self):
        return "Unresolved({})".format(self.what)


def _torch_tensor_from(rec):
    """A tensor written by a plain ``pickle.dump`` (the reference's terrain.pkl cache, dm_env.py:344-354, holds device tensors):
    ``_rebuild_tensor_v2(_load_from_bytes(blob), offset, size, stride, ...)``.  The blob is a legacy ``torch.save`` stream
    holding one storage; it is decoded by ``torch.load(weights_only=True)`` (restricted unpickler, data only) onto the CPU,
    and the view is rebuilt from the recorded offset / size / stride after a bounds check."""
    import io

    import torch
    st_rec, offset, size, stride = rec.args[0], rec.args[1], rec.args[2], rec.args[3]
    if not (isinstance(st_rec, Call) and _is_global(st_rec.func, ("torch.storage",), "_load_from_bytes")
            and isinstance(st_rec.args[0], (bytes, bytearray))):
        raise ValueError("unsupported tensor storage record")
    st = torch.load(io.BytesIO(bytes(st_rec.args[0])), weights_only=True, map_location="cpu")
    dtype = st.dtype
    ust = st._untyped_storage if hasattr(st, "_untyped_storage") else st
    size, stride = tuple(int(v) for v in size), tuple(int(v) for v in stride)
    last = int(offset) + sum((n - 1) * k for n, k in zip(size, stride)) if all(n > 0 for n in size) else -1
    if int(offset) < 0 or any(k < 0 for k in stride) or (last + 1) * torch.empty(0, dtype=dtype).element_size() > ust.nbytes():
        raise ValueError("tensor view outside its storage")
    return torch.empty(0, dtype=dtype).set_(ust, int(offset), size, stride).clone()


def materialize(rec, strict=False):
    """Turn the inert tree into plain python: numpy arrays/scalars, dicts, lists.

    ``Obj`` records (e.g. ``util.terrain_util.SubTerrain``) become
    ``{"__class__": "module.Name", **state}`` dicts.  Records that are not numpy data are
    left as ``Unresolved`` (or raise when ``strict``).
    """
    if isinstance(rec, Call):
        if _is_global(rec.func, ("numpy.core.multiarray", "numpy._core.multiarray"), "_reconstruct"):
            # state = (version, shape, dtype, is_fortran, rawbytes)
            _ver, shape, dt, fortran, raw = rec.state
            dtype = _dtype_from(dt)
            if not isinstance(raw, (bytes, bytearray)):
                raise ValueError("object arrays are not supported")
            arr = np.frombuffer(bytes(raw), dtype=dtype).copy()
            return arr.reshape(shape, order="F" if fortran else "C")
        if _is_global(rec.func, ("numpy.core.multiarray", "numpy._core.multiarray"), "scalar"):
            dtype = _dtype_from(rec.args[0])
            return np.frombuffer(bytes(rec.args[1]), dtype=dtype)[0]
        if _is_global(rec.func, ("torch._utils",), "_rebuild_tensor_v2"):
            return _torch_tensor_from(rec)
        if _is_global(rec.func, ("torch._utils",), "_rebuild_parameter"):
            return materialize(rec.args[0], strict)
        if strict:
            raise ValueError("refusing to evaluate {!r}".format(rec.func))
        return Unresolved(repr(rec.func))
    if isinstance(rec, Obj):
        out = OrderedDict()
        out["__class__"] = "{}.{}".format(rec.cls.module, rec.cls.name)
        state = rec.state if rec.state is not None else {}
        for k, v in state.items():
            out[k] = materialize(v, strict)
        return out
    if isinstance(rec, OrderedDict):
        return OrderedDict((k, materialize(v, strict)) for k, v in rec.items())
    if isinstance(rec, list):
        return [materialize(v, strict) for v in rec]
    if isinstance(rec, tuple):
        return tuple(materialize(v, strict) for v in rec)
    if isinstance(rec, Global):
        if strict:
            raise ValueError("bare global {!r} in data".format(rec))
        return Unresolved(repr(rec))
    return rec


def load_motion_file_safe(path):
    """Read a PARC motion pickle as plain data (numpy arrays + dicts); executes nothing."""
    with open(path, "rb") as f:
        data = f.read()
    return materialize(load_inert(data))


def load_executing(path):
    """The ONE place in this package that hands a file to ``pickle.load`` (as the reference does everywhere, e.g. anim/motion_lib.py:240):
    reached only through the explicit ``unsafe_pickle`` opt-ins of MotionLib / DeepMimicEnv / motion_edit_lib.load_motion_file, for files
    the user wrote themselves.  Never used on the files that ship with the reference."""
    import pickle
    from . import terrain_util
    # the files name their classes by the reference's module paths (util.terrain_util.SubTerrain ...): resolve those to this package for
    # the duration of the load, whether or not install_reference_aliases() was called - importing the package registers nothing
    with terrain_util.reference_pickle_path(), open(path, "rb") as f:
        return pickle.load(f)
