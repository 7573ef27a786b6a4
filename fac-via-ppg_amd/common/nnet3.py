"""Kaldi nnet3 "raw" acoustic models without Kaldi: reader, writer and the computation plan of a TDNN.

The reference gets its phonetic posteriorgrams from PyKaldi (absent here): ``decode.read_nnet3_model`` reads
``data/am/final.raw`` (decode.py:23-38) and ``compute_full_ppg`` runs it through ``nnet3.DecodableNnetSimple``
(compute_ppg.py:42-70).  The model file itself is NOT shipped by the reference (.MISSING_LARGE_BLOBS), and no
Kaldi-written nnet3 file exists in this image to check a reader against -- so everything in this module follows Kaldi's
published source (nnet3/nnet-nnet.cc ``Nnet::Read/Write``, nnet-simple-component.cc, nnet-normalize-component.cc,
nnet-descriptor.cc) as restated here, is exercised on models written by this module's own writer, and is
**parity-unpinned** at the Kaldi boundary.

File grammar (binary: ``\\0B`` first; tokens are ``<Name>`` + space; basic types carry a size byte):

    <Nnet3> \\n
    input-node name=input dim=40 \\n
    component-node name=tdnn1.affine component=tdnn1.affine input=Append(Offset(input, -2), ..., Offset(input, 2)) \\n
    ... one line per node ...   output-node name=output input=output.log-softmax objective=linear \\n
    \\n                                                  (blank line ends the config section)
    <NumComponents> n
    n x ( <ComponentName> name <Type> ...fields... </Type> )
    </Nnet3>

Supported components (what nnet3 TDNN recipes are made of): NaturalGradientAffineComponent, AffineComponent,
FixedAffineComponent (the LDA-like input layer), RectifiedLinearComponent, BatchNormComponent (test mode),
NormalizeComponent ("renorm"), SoftmaxComponent, LogSoftmaxComponent, NoOpComponent.  Descriptors: node names,
``Offset(x, t)``, ``Append(...)``, ``Sum(a, b)``, ``Scale(s, x)``.
"""
import re
import struct

import numpy as np


class Nnet3FormatError(ValueError):
    pass


INT_KEYS = {"Dim", "InputDim", "OutputDim", "BlockDim", "RankIn", "RankOut", "UpdatePeriod", "Rank", "NumComponents"}
# Kaldi keeps these counters as doubles (NonlinearComponent::num_dims_self_repaired_ / num_dims_processed_): 8-byte values in
# binary mode, possibly "1.2e+06" in text mode
DOUBLE_KEYS = {"Count", "OderivCount", "NumDimsSelfRepaired", "NumDimsProcessed"}
BOOL_KEYS = {"IsGradient", "TestMode", "AddLogStddev", "UseNaturalGradient"}
AFFINE_TYPES = ("NaturalGradientAffineComponent", "AffineComponent", "FixedAffineComponent")
NONLINEAR_TYPES = ("RectifiedLinearComponent", "SoftmaxComponent", "LogSoftmaxComponent", "NoOpComponent")


class Component(object):
    def __init__(self, type_, fields=None):
        self.type = type_
        self.fields = dict(fields or {})

    def __repr__(self):
        return "<%s %s>" % (self.type, {k: (v.shape if isinstance(v, np.ndarray) else v) for k, v in self.fields.items()})

    # ---- the parameters the computation needs
    @property
    def linear(self):
        return self.fields.get("LinearParams")

    @property
    def bias(self):
        return self.fields.get("BiasParams")

    def dims(self):
        """(input dim, output dim)"""
        if self.type in AFFINE_TYPES:
            return self.linear.shape[1], self.linear.shape[0]
        d = int(self.fields["Dim"]) if "Dim" in self.fields else int(self.fields["InputDim"])
        if self.type == "NormalizeComponent" and self.fields.get("AddLogStddev"):
            return d, d + 1
        return d, d


# --------------------------------------------------------------------------------------------- descriptors
def parse_descriptor(text):
    """'Append(Offset(a, -1), a, Offset(a, 1))' -> nested tuples ('Append', [...]) / ('Offset', x, t) / ('Sum', a, b) /
    ('Scale', s, x) / ('node', name)."""
    toks = re.findall(r"[A-Za-z_][\w.\-]*|-?\d+\.?\d*(?:[eE][-+]?\d+)?|[(),]", text)
    pos = [0]

    def expr():
        t = toks[pos[0]]
        pos[0] += 1
        if pos[0] < len(toks) and toks[pos[0]] == "(":
            pos[0] += 1
            args = []
            while toks[pos[0]] != ")":
                if toks[pos[0]] == ",":
                    pos[0] += 1
                    continue
                args.append(expr())
            pos[0] += 1
            if t == "Append":
                return ("Append", args)
            if t == "Offset":
                return ("Offset", args[0], int(args[1][1]))
            if t == "Sum":
                return ("Sum", args[0], args[1])
            if t == "Scale":
                return ("Scale", float(args[0][1]), args[1])
            raise Nnet3FormatError("descriptor function %r is not supported (in %r)" % (t, text))
        return ("node", t)

    out = expr()
    if pos[0] != len(toks):
        raise Nnet3FormatError("trailing tokens in descriptor %r" % text)
    return out


def descriptor_terms(d):
    """Flatten a descriptor made of Append / Offset over single nodes into [(node, offset)], in Append order; None if it
    also uses Sum / Scale."""
    kind = d[0]
    if kind == "node":
        return [(d[1], 0)]
    if kind == "Offset":
        inner = descriptor_terms(d[1])
        return None if inner is None else [(n, o + d[2]) for n, o in inner]
    if kind == "Append":
        out = []
        for a in d[1]:
            t = descriptor_terms(a)
            if t is None:
                return None
            out += t
        return out
    return None


def descriptor_text(d):
    kind = d[0]
    if kind == "node":
        return d[1]
    if kind == "Offset":
        return "Offset(%s, %d)" % (descriptor_text(d[1]), d[2])
    if kind == "Append":
        return "Append(%s)" % ", ".join(descriptor_text(a) for a in d[1])
    if kind == "Sum":
        return "Sum(%s, %s)" % (descriptor_text(d[1]), descriptor_text(d[2]))
    return "Scale(%r, %s)" % (d[1], descriptor_text(d[2]))


# --------------------------------------------------------------------------------------------- the network
class Nnet(object):
    """nodes: ordered list of dicts {kind: 'input' | 'component' | 'output', name, dim (input), component, input
    (descriptor)}; components: {name: Component}."""

    def __init__(self, nodes, components):
        self.nodes, self.components = nodes, components
        self.by_name = {n["name"]: n for n in nodes}

    def input_dim(self, name="input"):
        return int(self.by_name[name]["dim"])

    def node_dim(self, name):
        n = self.by_name[name]
        if n["kind"] == "input":
            return int(n["dim"])
        if n["kind"] == "component":
            return self.components[n["component"]].dims()[1]
        return self.descriptor_dim(n["input"])

    def descriptor_dim(self, d):
        if d[0] == "node":
            return self.node_dim(d[1])
        if d[0] == "Offset":
            return self.descriptor_dim(d[1])
        if d[0] == "Append":
            return sum(self.descriptor_dim(a) for a in d[1])
        if d[0] == "Sum":
            return self.descriptor_dim(d[1])
        return self.descriptor_dim(d[2])

    def output_dim(self, name="output"):
        return self.node_dim(name)

    def context(self, name="output"):
        """(left, right) input frames the node `name` at frame t depends on beyond t (ComputeSimpleNnetContext)."""
        memo = {}

        def span(node):
            if node in memo:
                return memo[node]
            n = self.by_name[node]
            r = (0, 0) if n["kind"] == "input" else dspan(n["input"])
            memo[node] = r
            return r

        def dspan(d):
            if d[0] == "node":
                return span(d[1])
            if d[0] == "Offset":
                lo, hi = dspan(d[1])
                return lo + d[2], hi + d[2]
            parts = [dspan(a) for a in (d[1] if d[0] == "Append" else (d[1], d[2]) if d[0] == "Sum" else (d[2],))]
            return min(p[0] for p in parts), max(p[1] for p in parts)

        lo, hi = span(name)
        return -lo, hi


# --------------------------------------------------------------------------------------------- reading
class _Stream(object):
    def __init__(self, buf, path):
        self.buf, self.path = buf, path
        self.binary = buf[:2] == b"\x00B"
        self.pos = 2 if self.binary else 0

    def fail(self, msg):
        raise Nnet3FormatError("%s: %s (at byte %d)" % (self.path, msg, self.pos))

    def skip_ws(self):
        while self.pos < len(self.buf) and self.buf[self.pos:self.pos + 1] in (b" ", b"\n", b"\t", b"\r"):
            self.pos += 1

    def peek(self, n=1):
        return self.buf[self.pos:self.pos + n]

    def token(self):
        self.skip_ws()
        end = self.pos
        while end < len(self.buf) and self.buf[end:end + 1] not in (b" ", b"\n", b"\t"):
            end += 1
        tok = self.buf[self.pos:end].decode("latin-1")
        self.pos = min(len(self.buf), end + 1)
        return tok

    def expect(self, tok):
        got = self.token()
        if got != tok:
            self.fail("expected %r, got %r" % (tok, got))

    def line(self):
        end = self.buf.index(b"\n", self.pos)
        s = self.buf[self.pos:end].decode("latin-1")
        self.pos = end + 1
        return s

    def _floats(self, n, double):
        dt = "<f8" if double else "<f4"
        nb = n * (8 if double else 4)
        if self.pos + nb > len(self.buf):
            self.fail("truncated vector / matrix")
        out = np.frombuffer(self.buf, dtype=dt, count=n, offset=self.pos).astype(np.float32)
        self.pos += nb
        return out

    def _bin_int(self):
        if self.buf[self.pos] != 4:
            self.fail("expected a 4-byte integer")
        v = struct.unpack_from("<i", self.buf, self.pos + 1)[0]
        self.pos += 5
        return v

    def value(self, key):
        """The value that follows token <key>, typed by what the stream says (and, for 4-byte basics, by the key)."""
        if self.binary:
            head = self.peek(3)
            if head in (b"FV ", b"DV "):
                self.pos += 3
                return self._floats(self._bin_int(), head == b"DV ")
            if head in (b"FM ", b"DM "):
                self.pos += 3
                r, c = self._bin_int(), self._bin_int()
                return self._floats(r * c, head == b"DM ").reshape(r, c)
            if head[:2] == b"CM":
                self.fail("compressed matrices are not supported")
            b0 = self.buf[self.pos]
            if b0 in (4, 8) or b0 in (252, 248):   # size byte (negative = unsigned)
                size = b0 if b0 < 128 else 256 - b0
                raw = self.buf[self.pos + 1:self.pos + 1 + size]
                if len(raw) < size:
                    self.fail("<%s>: the file ends inside a %d-byte value" % (key, size))
                self.pos += 1 + size
                if size == 8:
                    return struct.unpack("<d", raw)[0]
                return struct.unpack("<i", raw)[0] if key in INT_KEYS else struct.unpack("<f", raw)[0]
            if key in BOOL_KEYS and head[:1] in (b"T", b"F"):
                self.pos += 1
                return head[:1] == b"T"
            return None                               # a flag-like token without a value
        self.skip_ws()
        if self.peek() == b"[":
            self.pos += 1
            end = self.buf.index(b"]", self.pos)
            rows = [r.split() for r in self.buf[self.pos:end].decode().split("\n") if r.strip()]
            self.pos = end + 1
            arr = np.array([[float(x) for x in r] for r in rows], dtype=np.float32)
            return arr[0] if arr.shape[0] == 1 and key not in ("LinearParams", "Params") else arr
        if self.peek() == b"<":
            return None
        tok = self.token()
        if key in BOOL_KEYS:
            return tok == "T"
        try:
            return int(tok) if key in INT_KEYS else float(tok)
        except ValueError:
            self.fail("<%s>: %r is not a%s" % (key, tok, "n integer" if key in INT_KEYS else " number"))


def _read_component(st):
    open_tok = st.token()
    if not (open_tok.startswith("<") and open_tok.endswith(">")):
        st.fail("expected a component type, got %r" % open_tok)
    type_ = open_tok[1:-1]
    close = "</%s>" % type_
    fields = {}
    while True:
        tok = st.token()
        if tok == close:
            break
        if not (tok.startswith("<") and tok.endswith(">")):
            st.fail("expected a field token inside %s, got %r" % (type_, tok))
        fields[tok[1:-1]] = st.value(tok[1:-1])
    if type_ in AFFINE_TYPES:
        if type_ == "FixedAffineComponent" and "LinearParams" not in fields and "Params" in fields:
            p = fields["Params"]                   # [out, in + 1]: last column is the bias
            fields["LinearParams"], fields["BiasParams"] = p[:, :-1].copy(), p[:, -1].copy()
        if fields.get("LinearParams") is None or fields.get("BiasParams") is None:
            st.fail("%s without <LinearParams> / <BiasParams>" % type_)
    elif type_ not in NONLINEAR_TYPES + ("BatchNormComponent", "NormalizeComponent"):
        st.fail("component type %s is not supported" % type_)
    return Component(type_, fields)


def _parse_config_line(line):
    kind, _, rest = line.strip().partition(" ")
    kv = {}
    for m in re.finditer(r"(\w[\w-]*)=((?:[^\s(]+\([^=]*\))|\S+)(?=\s+\w[\w-]*=|\s*$)", rest):
        kv[m.group(1)] = m.group(2).strip()
    if kind == "input-node":
        return {"kind": "input", "name": kv["name"], "dim": int(kv["dim"])}
    if kind == "component-node":
        return {"kind": "component", "name": kv["name"], "component": kv["component"], "input": parse_descriptor(kv["input"])}
    if kind == "output-node":
        return {"kind": "output", "name": kv["name"], "input": parse_descriptor(kv["input"]), "objective": kv.get("objective", "linear")}
    raise Nnet3FormatError("config line %r is not supported (dim-range-node etc.)" % line)


def read_nnet3(path):
    """A raw nnet3 model file (binary or text) -> Nnet."""
    with open(path, "rb") as f:
        st = _Stream(f.read(), path)
    st.expect("<Nnet3>")
    if st.peek() == b"\n":
        st.pos += 1
    nodes = []
    while True:
        line = st.line()
        if not line.strip():
            break
        nodes.append(_parse_config_line(line))
    st.expect("<NumComponents>")
    n = st.value("NumComponents")
    comps = {}
    for _ in range(int(n)):
        st.expect("<ComponentName>")
        name = st.token()
        comps[name] = _read_component(st)
    st.expect("</Nnet3>")
    for node in nodes:
        if node["kind"] == "component" and node["component"] not in comps:
            raise Nnet3FormatError("%s: node %s refers to a missing component %s" % (path, node["name"], node["component"]))
    return Nnet(nodes, comps)


# --------------------------------------------------------------------------------------------- writing
def _w_token(out, tok):
    out.append(tok.encode() + b" ")


def _w_value(out, key, v, binary):
    if isinstance(v, np.ndarray):
        v = v.astype("<f4")
        if binary:
            if v.ndim == 1:
                out.append(b"FV " + b"\x04" + struct.pack("<i", v.shape[0]) + v.tobytes())
            else:
                out.append(b"FM " + b"\x04" + struct.pack("<i", v.shape[0]) + b"\x04" + struct.pack("<i", v.shape[1]) + v.tobytes())
        else:
            rows = v.reshape(1, -1) if v.ndim == 1 else v
            out.append((" [\n  " if v.ndim == 2 else " [ ").encode() + "\n  ".join(" ".join(repr(float(x)) for x in r) for r in rows).encode() + b" ]\n")
    elif isinstance(v, bool):
        out.append(b"T" if v else b"F")
        if not binary:
            out.append(b" ")
    elif v is None:
        pass
    elif binary:
        if key in DOUBLE_KEYS:
            out.append(b"\x08" + struct.pack("<d", float(v)))
        elif key in INT_KEYS:
            out.append(b"\x04" + struct.pack("<i", int(v)))
        else:
            out.append(b"\x04" + struct.pack("<f", float(v)))
    else:
        out.append((("%d " % v) if key in INT_KEYS else ("%r " % float(v))).encode())


def write_nnet3(path, nnet, binary=True):
    """Write `nnet` in the grammar read_nnet3 reads (the module docstring's restatement of Nnet::Write)."""
    out = [b"\x00B"] if binary else []
    _w_token(out, "<Nnet3>")
    out.append(b"\n")
    for n in nnet.nodes:
        if n["kind"] == "input":
            out.append(("input-node name=%s dim=%d\n" % (n["name"], n["dim"])).encode())
        elif n["kind"] == "component":
            out.append(("component-node name=%s component=%s input=%s\n" % (n["name"], n["component"], descriptor_text(n["input"]))).encode())
        else:
            out.append(("output-node name=%s input=%s objective=%s\n" % (n["name"], descriptor_text(n["input"]), n.get("objective", "linear"))).encode())
    out.append(b"\n")
    _w_token(out, "<NumComponents>")
    _w_value(out, "NumComponents", len(nnet.components), binary)
    if not binary:
        out.append(b"\n")
    for name, c in nnet.components.items():
        _w_token(out, "<ComponentName>")
        _w_token(out, name)
        _w_token(out, "<%s>" % c.type)
        for k, v in c.fields.items():
            _w_token(out, "<%s>" % k)
            _w_value(out, k, v, binary)
        _w_token(out, "</%s>" % c.type)
        if not binary:
            out.append(b"\n")
    _w_token(out, "</Nnet3>")
    with open(path, "wb") as f:
        f.write(b"".join(out))


# --------------------------------------------------------------------------------------------- synthetic models
def synthetic_tdnn(input_dim=40, hidden=64, output_dim=96, splices=((-2, -1, 0, 1, 2), (-1, 2), (-3, 3), (0,)), norm="batchnorm",
                   output="softmax", seed=0, lda=True):
    """A TDNN of the shape nnet3's recipes build (xconfig `relu-batchnorm-layer` / `relu-renorm-layer` stacks): an
    optional FixedAffine 'lda' layer on the spliced input, then per splice an affine -> ReLU -> BatchNorm | Normalize, a
    final affine and (Log)Softmax.  Weights ~ N(0, 1/fan_in) from PCG64(seed); batch-norm statistics non-trivial."""
    g = np.random.Generator(np.random.PCG64(seed))
    nodes, comps = [{"kind": "input", "name": "input", "dim": input_dim}], {}
    prev, prev_dim = "input", input_dim

    def affine(name, type_, in_dim, out_dim, extra=None):
        f = {}
        if type_ != "FixedAffineComponent":
            f.update({"LearningRateFactor": 1.0, "MaxChange": 0.75, "LearningRate": 0.001})
        f["LinearParams"] = (g.standard_normal((out_dim, in_dim)) / np.sqrt(in_dim)).astype(np.float32)
        f["BiasParams"] = (0.1 * g.standard_normal(out_dim)).astype(np.float32)
        if type_ == "NaturalGradientAffineComponent":
            f.update({"RankIn": 20, "RankOut": 80, "UpdatePeriod": 4, "NumSamplesHistory": 2000.0, "Alpha": 4.0})
        comps[name] = Component(type_, f)

    def desc(src, offs):
        terms = [("node", src) if o == 0 else ("Offset", ("node", src), o) for o in offs]
        return terms[0] if len(terms) == 1 else ("Append", terms)

    for li, offs in enumerate(splices):
        in_dim = prev_dim * len(offs)
        if li == 0 and lda:
            affine("lda", "FixedAffineComponent", in_dim, in_dim)
            nodes.append({"kind": "component", "name": "lda", "component": "lda", "input": desc(prev, offs)})
            prev, offs = "lda", (0,)
            prev_dim = in_dim
            in_dim = prev_dim
        base = "tdnn%d" % (li + 1)
        affine(base + ".affine", "NaturalGradientAffineComponent", in_dim, hidden)
        nodes.append({"kind": "component", "name": base + ".affine", "component": base + ".affine", "input": desc(prev, offs)})
        comps[base + ".relu"] = Component("RectifiedLinearComponent", {"Dim": hidden, "ValueAvg": np.zeros(0, np.float32),
                                                                       "DerivAvg": np.zeros(0, np.float32), "Count": 0.0,
                                                                       "NumDimsSelfRepaired": 0.0, "NumDimsProcessed": 0.0})
        nodes.append({"kind": "component", "name": base + ".relu", "component": base + ".relu", "input": ("node", base + ".affine")})
        if norm == "batchnorm":
            comps[base + ".batchnorm"] = Component("BatchNormComponent", {
                "Dim": hidden, "BlockDim": hidden, "Epsilon": 0.001, "TargetRms": 1.0, "TestMode": False, "Count": 1000.0,
                "StatsMean": (0.4 + 0.2 * g.random(hidden)).astype(np.float32), "StatsVar": (0.3 + 0.5 * g.random(hidden)).astype(np.float32)})
            last = base + ".batchnorm"
        else:
            comps[base + ".renorm"] = Component("NormalizeComponent", {"Dim": hidden, "TargetRms": 1.0, "AddLogStddev": False})
            last = base + ".renorm"
        nodes.append({"kind": "component", "name": last, "component": last, "input": ("node", base + ".relu")})
        prev, prev_dim = last, hidden
    affine("output.affine", "NaturalGradientAffineComponent", prev_dim, output_dim)
    nodes.append({"kind": "component", "name": "output.affine", "component": "output.affine", "input": ("node", prev)})
    kind = "SoftmaxComponent" if output == "softmax" else "LogSoftmaxComponent"
    oname = "output.softmax" if output == "softmax" else "output.log-softmax"
    comps[oname] = Component(kind, {"Dim": output_dim, "ValueAvg": np.zeros(0, np.float32), "DerivAvg": np.zeros(0, np.float32), "Count": 0.0,
                                    "NumDimsSelfRepaired": 0.0, "NumDimsProcessed": 0.0})
    nodes.append({"kind": "component", "name": oname, "component": oname, "input": ("node", "output.affine")})
    nodes.append({"kind": "output", "name": "output", "input": ("node", oname), "objective": "linear"})
    return Nnet(nodes, comps)


# --------------------------------------------------------------------------------------------- computation plan
def plan_layers(nnet, output="output"):
    """The network as a chain of fused layers for the HIP path (csrc/facppg_tdnn.hip):

        [{'W': [out, n_taps * in] fp64, 'b': [out] fp64, 'first': offset of tap 0, 'dil': tap spacing, 'taps': n,
          'act': 'none' | 'relu', 'renorm': target_rms or None}, ...], final ('softmax' | 'log-softmax' | 'none')

    Affine layers read Append(Offset(x, o_j)) of ONE source; offsets are embedded in the uniform grid first + j * dil
    (missing grid points get zero weights).  Test-mode BatchNorm is a per-channel scale and shift AFTER the ReLU: it is
    folded into the consuming affine layer's weights and bias (as nnet3's own ``collapse_model`` does, compute_ppg.py:57),
    consecutive affine layers without a nonlinearity in between (the LDA layer) are multiplied together, all in fp64."""
    order = []
    node = nnet.by_name[output]
    terms = descriptor_terms(node["input"])
    if terms is None or len(terms) != 1 or terms[0][1] != 0:
        raise Nnet3FormatError("output node must read one component node")
    cur = terms[0][0]
    while nnet.by_name[cur]["kind"] != "input":
        n = nnet.by_name[cur]
        t = descriptor_terms(n["input"])
        if t is None or len({src for src, _ in t}) != 1:
            raise Nnet3FormatError("node %s: only Append / Offset of ONE source node is supported on the HIP path" % cur)
        order.append((nnet.components[n["component"]], [o for _, o in t]))
        cur = t[0][0]
    order.reverse()
    layers, final = [], "none"
    pending_scale = pending_shift = None           # per-channel y = x * scale + shift waiting for the next affine
    for comp, offs in order:
        if comp.type in AFFINE_TYPES:
            W, b = comp.linear.astype(np.float64), comp.bias.astype(np.float64)
            in_dim = W.shape[1] // len(offs)
            srt = sorted(offs)
            if srt != list(offs):
                raise Nnet3FormatError("Append offsets must be increasing")
            dil = int(np.gcd.reduce(np.diff(srt))) if len(srt) > 1 else 1
            taps = (srt[-1] - srt[0]) // dil + 1
            Wg = np.zeros((W.shape[0], taps, in_dim))
            for j, o in enumerate(offs):
                Wg[:, (o - srt[0]) // dil, :] = W[:, j * in_dim:(j + 1) * in_dim]
            if pending_scale is not None:
                b = b + np.einsum("otc,c->o", Wg, pending_shift)
                Wg = Wg * pending_scale[None, None, :]
                pending_scale = pending_shift = None
            prev = layers[-1] if layers else None
            # (an affine reading ONE frame at offset 0: with Offset(prev, k), k != 0, the product would have to shift prev's taps)
            if prev is not None and prev["act"] == "none" and prev["renorm"] is None and taps == 1 and srt[0] == 0:
                prev["b"] = Wg[:, 0, :] @ prev["b"] + b            # affine after affine: one matrix
                prev["W"] = np.einsum("oc,ctk->otk", Wg[:, 0, :], prev["W"].reshape(prev["W"].shape[0], prev["taps"], -1)).reshape(W.shape[0], -1)
            else:
                layers.append({"W": Wg.reshape(W.shape[0], -1), "b": b, "first": srt[0], "dil": dil, "taps": taps, "act": "none", "renorm": None})
            continue
        if offs != [0]:
            raise Nnet3FormatError("a %s reading a spliced input is not supported" % comp.type)
        if comp.type == "RectifiedLinearComponent":
            if not layers or layers[-1]["act"] != "none" or layers[-1]["renorm"] is not None or pending_scale is not None:
                raise Nnet3FormatError("ReLU must directly follow an affine component")
            layers[-1]["act"] = "relu"
        elif comp.type == "BatchNormComponent":
            f = comp.fields
            if int(f.get("BlockDim", f["Dim"])) != int(f["Dim"]):
                raise Nnet3FormatError("BatchNorm with block-dim != dim is not supported")
            mean, var = f["StatsMean"].astype(np.float64), f["StatsVar"].astype(np.float64)
            scale = float(f.get("TargetRms", 1.0)) / np.sqrt(var + float(f.get("Epsilon", 1e-3)))
            s0 = pending_scale if pending_scale is not None else 1.0
            t0 = pending_shift if pending_shift is not None else 0.0
            pending_scale, pending_shift = s0 * scale, t0 * scale - mean * scale
        elif comp.type == "NormalizeComponent":
            if comp.fields.get("AddLogStddev"):
                raise Nnet3FormatError("NormalizeComponent with add-log-stddev is not supported")
            if pending_scale is not None or not layers or layers[-1]["renorm"] is not None:
                raise Nnet3FormatError("NormalizeComponent must follow an affine (+ ReLU)")
            layers[-1]["renorm"] = float(comp.fields.get("TargetRms", 1.0))
        elif comp.type in ("SoftmaxComponent", "LogSoftmaxComponent"):
            final = "softmax" if comp.type == "SoftmaxComponent" else "log-softmax"
        elif comp.type != "NoOpComponent":
            raise Nnet3FormatError("component %s is not supported on the HIP path" % comp.type)
    if pending_scale is not None:
        raise Nnet3FormatError("a trailing BatchNorm with no affine after it is not supported")
    return layers, final
