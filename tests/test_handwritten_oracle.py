"""The oracle against closed-form known answers of softmax / crossEntropy / adam / 4-D conv2 /
maxpool2 + customGrad, on programs written by hand in the kernel-description grammar (no DSL
mirror involved) — tests/golden/make_handwritten.py."""
import pytest

import handwritten

CASES = handwritten.load()


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_the_closed_forms(name):
    from oracle import kd
    case = CASES[name]
    ref = kd.Model(case["text"])

    def set_param(t, v):
        ref.params[t][...] = v

    def set_epoch(e):
        ref.epoch = e

    handwritten.check(case, ref, set_param, lambda t: ref.params[t], lambda t: ref.caches[t], set_epoch)


@pytest.mark.parametrize("name", sorted(CASES))
def test_handwritten_text_equals_what_the_dsl_mirror_emits_semantically(name):
    """The DSL mirror's own text for the same layers, run on the oracle, gives the same numbers: a
    front-end bug outside the 37 pinned reference programs would show here."""
    import numpy as np
    import refcases
    from exprgrad_amd import dsl, layers
    from oracle import kd
    case = CASES[name]
    arr = handwritten.arr
    if name == "conv2_4d":
        ref = kd.Model(refcases.program_text(refcases.conv2_bench()))
        got = ref.call("conv2", {k: arr(v).astype(np.float32) for k, v in case["inputs"].items()})
        assert np.array_equal(got, arr(case["calls"][0]["expect"]))
    elif name == "maxpool2_grad":
        net = layers.maxpool2(dsl.input("x")).target("pool")
        ref = kd.Model(refcases.program_text([net]))
        got = ref.call("pool", {"x": arr(case["inputs"]["x"]).astype(np.float32)})
        assert np.array_equal(got, arr(case["calls"][0]["expect"]))
    elif name == "xor_from_scratch":
        from exprgrad_amd import examples
        ref = kd.Model(refcases.program_text(examples.xor_from_scratch()))
        for tid, spec in case["params"].items():        # (the mirror allocates tensors like parser.nim does: same ids)
            ref.params[int(tid)][...] = arr(spec).astype(np.float32)
        ins = {k: arr(v).astype(np.float32) for k, v in case["inputs"].items()}
        assert handwritten.close(ref.call("predict", {"x": ins["x"]}), arr(case["calls"][0]["expect"]), 1e-6)
        assert handwritten.close(ref.call("loss", ins), arr(case["calls"][1]["expect"]), 1e-6)
        ref.apply("train", ins)
        for tid, spec in case["expect_params"].items():
            assert handwritten.close(ref.params[int(tid)], arr(spec), 1e-6), tid
    elif name == "pool_chain_adam":
        from exprgrad_amd import examples
        ref = kd.Model(refcases.program_text(examples.pool_chain_adam()))
        ref.params[1][...] = arr(case["params"]["1"]).astype(np.float32)
        ref.epoch = 1
        ins = {k: arr(v).astype(np.float32) for k, v in case["inputs"].items()}
        assert handwritten.close(ref.call("predict", {"x": ins["x"]}), arr(case["calls"][0]["expect"]), case["tol"])
        assert handwritten.close(ref.call("loss", ins), arr(case["calls"][1]["expect"]), case["tol"])
        ref.apply("fit", ins)
        assert handwritten.close(ref.params[1], arr(case["expect_params"]["1"]), case["tol"])
    elif name == "softmax_xent":
        net = layers.softmax(dsl.input("z")).target("predict")
        net = layers.cross_entropy(net, dsl.input("y")).target("loss")
        ref = kd.Model(refcases.program_text([net]))
        ins = {"z": arr(case["params"]["1"]).astype(np.float32), "y": arr(case["inputs"]["y"]).astype(np.float32)}
        assert handwritten.close(ref.call("predict", {"z": ins["z"]}), arr(case["calls"][0]["expect"]), 1e-6)
        assert handwritten.close(ref.call("loss", ins), arr(case["calls"][1]["expect"]), 1e-6)
    else:
        it = dsl.iters("it")
        p = dsl.param([3], name="p")
        loss = dsl.Fun()
        loss[0] += dsl.sq(p.raw[it] - dsl.input("t").raw[it])
        ref = kd.Model(refcases.program_text([loss.target("loss").backprop(layers.adam(0.5)).target("train")]))
        tid = sorted(ref.params)[0]
        ref.params[tid][...] = arr(case["params"]["1"]).astype(np.float32)
        ref.epoch = 1
        ref.apply("train", {"t": arr(case["inputs"]["t"]).astype(np.float32)})
        assert handwritten.close(ref.params[tid], arr(case["expect_params"]["1"]), 1e-5)


def _normalised(text, keep_register_file_sizes=False):
    """Kernel-description text up to what is not information: comments, indentation, loop labels, the spelling of
    float literals, register-file sizes (a register that no line mentions) and the order of the targets."""
    targets, head, cur = {}, [], None
    for line in text.splitlines():
        t = line.split()
        if not t or t[0].startswith("#"):
            continue
        if t[0] == "loop":
            t[2] = "_"
        elif t[0] == "kernel" and not keep_register_file_sizes:
            t[1] = "_"
        elif t[0] == "ins" and t[1] == "scalar":
            t[-1] = repr(float(t[-1]))
        elif t[0] == "tensor" and t[2] in ("param", "random"):
            t[-2:] = [repr(float(t[-2])), repr(float(t[-1]))]
        if t[0] == "target":
            cur = targets.setdefault(t[1], [t])
        elif t[0] == "endtarget":
            cur.append(t)
            cur = None
        else:
            (head if cur is None else cur).append(t)
    return head, [targets[k] for k in sorted(targets)]


def test_the_handwritten_xor_text_and_the_dsl_mirror_agree_line_by_line():
    """tests/golden/handwritten/xor_from_scratch.kd was derived by hand from parser.nim (tensor ids, kernel order,
    register numbers — see its header); exprgrad_amd/dsl.py restates the same front-end in Python.  Two independent
    readings of the reference must give the same program, line for line."""
    import refcases
    from exprgrad_amd import examples
    mirror = refcases.program_text(examples.xor_from_scratch())
    assert _normalised(CASES["xor_from_scratch"]["text"]) == _normalised(mirror)


def test_the_handwritten_pool_chain_text_and_the_dsl_mirror_agree_line_by_line():
    """reshape -> conv2 -> maxpool2 (customGrad) -> mse -> adam: GenReshape, six iterators with `y + dy` indices, `y * 2 + 1`
    and `y div 2`, shape(), caches, epoch() — tests/golden/handwritten/pool_chain_adam.kd was derived by hand from
    parser.nim / passes.nim / toKd (see its header).  Here the register-file sizes are compared as well: the mirror must
    allocate a register for every node of an index expression exactly where build() does (parser.nim:159-217)."""
    import refcases
    from exprgrad_amd import examples
    mirror = refcases.program_text(examples.pool_chain_adam())
    want, got = _normalised(CASES["pool_chain_adam"]["text"], True), _normalised(mirror, True)
    assert want[0] == got[0]
    for a, b in zip(want[1], got[1]):
        assert a == b, next((x, y) for x, y in zip(a, b) if x != y)
    assert len(want[1]) == len(got[1])


def test_host_valued_index_instructions_may_be_written_as_idx():
    """`idx` lines that do not depend on an iterator (a literal, shape(), len(), epoch()) are host values whichever
    keyword a producer used for them (VERDICT r3 missing #6: shape()/len()/epoch() inside a computed index): the text with
    the literals of `y div 2` as "idx" instead of "setup" runs to the same closed form."""
    import numpy as np
    from oracle import kd
    case = CASES["pool_chain_adam"]
    text = case["text"].replace("  setup index ", "  idx index ")
    assert text != case["text"]
    ref = kd.Model(text)
    ref.params[1][...] = handwritten.arr(case["params"]["1"]).astype(np.float32)
    ref.epoch = 1
    ins = {k: handwritten.arr(v).astype(np.float32) for k, v in case["inputs"].items()}
    ref.apply("fit", ins)
    assert handwritten.close(ref.params[1], handwritten.arr(case["expect_params"]["1"]), case["tol"])
