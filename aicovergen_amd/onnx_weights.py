"""MDX-Net parameters out of the distributed `.onnx` graphs (reference: onnxruntime.InferenceSession(model_path),
src/mdx.py:74-77; the graph is run at :193).  This module does NOT run the graph: it reads the ModelProto with a minimal
protobuf wire-format decoder (no `onnx` / `protobuf` dependency), walks the nodes in their (topological) file order and
maps the Conv / ConvTranspose / MatMul / BatchNormalization operands onto the parameter names of the published kuielab
`ConvTDFNet` (first_conv, ds_dense.i.tfc.H.j / .tdf, ds.i, mid_dense, us.i, us_dense.i, final_conv) that
`aicovergen_amd.mdx_net.ConvTDFNet` consumes.

What torch.onnx.export does to that network in eval mode (checked against graphs exported here, tests/test_onnx_weights.py):
  * Conv2d + BatchNorm2d are fused into one Conv with a bias ("onnx::Conv_NNN" initializers): the BatchNorm is emitted as
    identity statistics for those layers;
  * Linear on a 4-D tensor becomes MatMul (weight transposed to (in, out), "onnx::MatMul_NNN") [+ Add for a bias], the
    BatchNorm2d after it stays a BatchNormalization node;
  * ConvTranspose2d keeps its BatchNormalization node.
PARITY: pinned against torch.onnx.export of the restated network only -- no published UVR `.onnx` file is available offline.
"""
import struct

import numpy as np
import torch

# ------------------------------------------------------------------------------------------------------------------
# protobuf wire format (https://protobuf.dev/programming-guides/encoding/): just what ModelProto / GraphProto /
# NodeProto / TensorProto / AttributeProto need
# ------------------------------------------------------------------------------------------------------------------


def _varint(buf, pos):
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _fields(buf):
    """Yield (field number, wire type, value) of one message; length-delimited values are memoryview slices."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val = bytes(buf[pos:pos + 8])
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            val = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            val = bytes(buf[pos:pos + 4])
            pos += 4
        else:
            raise ValueError("onnx: unsupported protobuf wire type %d" % wt)
        yield fno, wt, val


def _packed_varints(val, wt):
    if wt == 0:
        return [val]
    out, pos = [], 0
    while pos < len(val):
        v, pos = _varint(val, pos)
        out.append(v)
    return out


def _signed(v):
    return v - (1 << 64) if v >= (1 << 63) else v


_DTYPES = {1: np.float32, 6: np.int32, 7: np.int64, 10: np.float16, 11: np.float64}


def _tensor(buf):
    """TensorProto -> (name, numpy array).  dims=1, data_type=2, float_data=4, int32_data=5, int64_data=7, name=8,
    raw_data=9, double_data=10."""
    dims, dtype, name, raw = [], 1, "", None
    floats, ints = [], []
    for fno, wt, val in _fields(buf):
        if fno == 1:
            dims += [_signed(v) for v in _packed_varints(val, wt)]
        elif fno == 2:
            dtype = val
        elif fno == 8:
            name = bytes(val).decode()
        elif fno == 9:
            raw = bytes(val)
        elif fno == 4:
            floats.append(bytes(val))
        elif fno in (5, 7):
            ints += [_signed(v) for v in _packed_varints(val, wt)]
    if dtype not in _DTYPES:
        raise ValueError("onnx: initializer %s has unsupported data_type %d" % (name, dtype))
    if raw is not None:
        arr = np.frombuffer(raw, dtype=_DTYPES[dtype])
    elif floats:
        arr = np.frombuffer(b"".join(floats), dtype=np.float32)
    else:
        arr = np.asarray(ints, dtype=_DTYPES[dtype])
    return name, arr.reshape(dims).copy()


def _attribute(buf):
    """AttributeProto -> (name, value): f=2, i=3, s=4, t=5, floats=7, ints=8."""
    name, val = "", None
    ints = []
    for fno, wt, v in _fields(buf):
        if fno == 1:
            name = bytes(v).decode()
        elif fno == 2:
            val = struct.unpack("<f", v)[0]
        elif fno == 3:
            val = _signed(v)
        elif fno == 4:
            val = bytes(v).decode(errors="replace")
        elif fno == 5:
            val = _tensor(v)[1]
        elif fno == 8:
            ints += [_signed(x) for x in _packed_varints(v, wt)]
    return name, (ints if ints else val)


def _node(buf):
    """NodeProto: input=1, output=2, name=3, op_type=4, attribute=5."""
    node = {"input": [], "output": [], "name": "", "op": "", "attr": {}}
    for fno, wt, v in _fields(buf):
        if fno == 1:
            node["input"].append(bytes(v).decode())
        elif fno == 2:
            node["output"].append(bytes(v).decode())
        elif fno == 3:
            node["name"] = bytes(v).decode()
        elif fno == 4:
            node["op"] = bytes(v).decode()
        elif fno == 5:
            k, a = _attribute(v)
            node["attr"][k] = a
    return node


def parse_model(path):
    """-> (nodes in file order, {initializer name: numpy array}).  ModelProto.graph = 7; GraphProto.node = 1,
    .initializer = 5.  Constant nodes and Identity aliases of initializers are folded into the initializer table."""
    data = memoryview(open(path, "rb").read())
    graph = None
    for fno, wt, v in _fields(data):
        if fno == 7 and wt == 2:
            graph = v
    if graph is None:
        raise ValueError("%s: no GraphProto found (not an ONNX ModelProto?)" % path)
    nodes, inits = [], {}
    for fno, wt, v in _fields(graph):
        if fno == 1:
            nodes.append(_node(v))
        elif fno == 5:
            name, arr = _tensor(v)
            inits[name] = arr
    for nd in nodes:  # file order is topological: aliases of aliases resolve
        if nd["op"] == "Constant" and "value" in nd["attr"]:
            inits[nd["output"][0]] = nd["attr"]["value"]
        elif nd["op"] == "Identity" and nd["input"][0] in inits:
            inits[nd["output"][0]] = inits[nd["input"][0]]  # the exporter de-duplicates equal initializers this way
    return nodes, inits


# ------------------------------------------------------------------------------------------------------------------
# graph walk -> kuielab ConvTDFNet state_dict
# ------------------------------------------------------------------------------------------------------------------
_BN_EPS = 1e-5


def _identity_bn(sd, name, c):
    # running_var + eps == 1 exactly: folding this BatchNorm is the identity
    sd[name + ".weight"] = torch.ones(c)
    sd[name + ".bias"] = torch.zeros(c)
    sd[name + ".running_mean"] = torch.zeros(c)
    sd[name + ".running_var"] = torch.full((c,), 1.0 - _BN_EPS)


def _collect_layers(nodes, inits):
    """Linear list of parameterised layers in execution order: ("conv" | "convT" | "linear", weight, bias, bn | None)
    where bn = (gamma, beta, mean, var, eps) when a BatchNormalization consumes the layer's output."""
    consumers = {}
    for nd in nodes:
        for i in nd["input"]:
            consumers.setdefault(i, []).append(nd)

    def follow(out, op):
        nxt = consumers.get(out, [])
        return nxt[0] if len(nxt) == 1 and nxt[0]["op"] == op else None

    layers = []
    for nd in nodes:
        op = nd["op"]
        if op in ("Conv", "ConvTranspose"):
            w = inits.get(nd["input"][1])
            if w is None:
                raise ValueError("onnx: %s %s has a non-constant weight" % (op, nd["name"]))
            b = inits.get(nd["input"][2]) if len(nd["input"]) > 2 else None
            out = nd["output"][0]
        elif op == "MatMul":
            w = inits.get(nd["input"][1])
            if w is None or w.ndim != 2:
                continue  # activation x activation product: not a parameterised layer
            w = np.ascontiguousarray(w.T)  # (in, out) -> nn.Linear's (out, in)
            b, out = None, nd["output"][0]
            add = follow(out, "Add")
            if add is not None:
                other = [i for i in add["input"] if i != out]
                if other and other[0] in inits and inits[other[0]].ndim == 1:
                    b, out = inits[other[0]], add["output"][0]
        elif op in ("Gemm", "Einsum", "ConvInteger", "MatMulInteger", "QLinearConv", "QLinearMatMul"):
            raise ValueError("onnx: parameterised node %r (%s) is not part of the TFC-TDF U-Net this reader maps "
                             "(Conv / ConvTranspose / MatMul [+ Add] [+ BatchNormalization])" % (nd["name"], op))
        else:
            continue
        bn = None
        bnode = follow(out, "BatchNormalization")
        if bnode is not None:
            g, be, mu, var = (inits[i] for i in bnode["input"][1:5])
            bn = (g, be, mu, var, float(bnode["attr"].get("epsilon", 1e-5)))
        else:
            # graph optimisers rewrite an eval-mode BatchNormalization as Mul(per-channel scale) [-> Add(per-channel shift)] with
            # (1, C, 1, 1) constants; taken as gamma = scale, beta = shift over unit statistics.  Anything else that is not a
            # plain activation between two parameterised layers is refused below rather than skipped.
            def chan_const(node):
                other = [i for i in node["input"] if i != cur]
                c = inits.get(other[0]) if other else None
                if c is None:
                    return None
                c = np.asarray(c)
                return c.reshape(-1) if c.ndim in (3, 4) and c.size == c.shape[-3] else None
            cur = out
            mul = follow(cur, "Mul")
            scale = chan_const(mul) if mul is not None else None
            if scale is not None:
                cur = mul["output"][0]
                add = follow(cur, "Add")
                shift = chan_const(add) if add is not None else None
                bn = (scale, shift if shift is not None else np.zeros_like(scale), np.zeros_like(scale),
                      np.full_like(scale, 1.0 - _BN_EPS), _BN_EPS)
        kind = {"Conv": "conv", "ConvTranspose": "convT", "MatMul": "linear"}[op]
        layers.append({"kind": kind, "w": w, "b": b, "bn": bn, "attr": nd["attr"]})
    return layers


def state_dict_from_layers(layers):
    """Assign the execution-ordered layers to the ConvTDFNet parameter names.  Structure (ConvTDFNet.forward):
    first_conv | n x [l x tfc conv, 2 x tdf linear, strided ds conv] | mid (l conv, 2 linear) |
    n x [us convT, l conv, 2 linear] | final_conv."""
    kinds = "".join({"conv": "c", "convT": "t", "linear": "m"}[la["kind"]] for la in layers)
    n = kinds.count("t")
    if n == 0 or not kinds.startswith("c") or not kinds.endswith("c"):
        raise ValueError("onnx: layer sequence %r is not a TFC-TDF U-Net" % kinds)
    first_block = kinds[1:kinds.index("m")]
    l = len(first_block)  # tfc convs before the first tdf linear
    expect = "c" + ("c" * l + "mm" + "c") * n + "c" * l + "mm" + ("t" + "c" * l + "mm") * n + "c"
    if kinds != expect:
        raise ValueError("onnx: layer sequence %r does not match n=%d, l=%d (%r)" % (kinds, n, l, expect))
    sd = {}

    def put(prefix, la, bn_name):
        w = torch.from_numpy(np.ascontiguousarray(la["w"])).float()
        cout = w.shape[1] if la["kind"] == "convT" else w.shape[0]
        sd[prefix + ".weight"] = w
        sd[prefix + ".bias"] = torch.from_numpy(la["b"].copy()).float() if la["b"] is not None else torch.zeros(cout)
        return cout

    def put_bn(name, la, c):
        if la["bn"] is None:
            _identity_bn(sd, name, c)
            return
        g, be, mu, var, eps = la["bn"]
        sd[name + ".weight"] = torch.from_numpy(g.copy()).float()
        sd[name + ".bias"] = torch.from_numpy(be.copy()).float()
        sd[name + ".running_mean"] = torch.from_numpy(mu.copy()).float()
        # the consumer folds with eps = 1e-5: keep var + eps equal to what the graph says
        sd[name + ".running_var"] = torch.from_numpy(var.copy()).float() + (eps - _BN_EPS)

    it = iter(layers)

    def conv_bn(name):
        la = next(it)
        c = put(name + ".0", la, name + ".1")
        put_bn(name + ".1", la, c)

    def tfc_tdf(name, c_channels):
        for j in range(l):
            conv_bn("%s.tfc.H.%d" % (name, j))
        for idx in (0, 3):
            la = next(it)
            put("%s.tdf.%d" % (name, idx), la, None)
            # BatchNorm2d over the channel axis of (B, C, T, F): C entries, not the linear's output width
            put_bn("%s.tdf.%d" % (name, idx + 1), la, c_channels)

    conv_bn("first_conv")
    c = sd["first_conv.0.weight"].shape[0]
    g = c
    for i in range(n):
        tfc_tdf("ds_dense.%d" % i, c)
        conv_bn("ds.%d" % i)
        c += g
    tfc_tdf("mid_dense", c)
    for i in range(n):
        conv_bn("us.%d" % i)
        c -= g
        tfc_tdf("us_dense.%d" % i, c)
    la = next(it)
    put("final_conv.0", la, None)
    return sd


def load_onnx_state_dict(path):
    nodes, inits = parse_model(path)
    return state_dict_from_layers(_collect_layers(nodes, inits))
