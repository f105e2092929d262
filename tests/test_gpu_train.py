"""GPU tests of the native OAA learner, the model container and the CLI pipeline.

The learner is this project's own VW-like rule (VW is absent: parity unpinned), so the checks are: (1) the HIP kernel
in `sequential` mode against the oracle's sequential restatement of the same rule (tolerance 2e-4 relative: GPU expf
differs from libm's in the last bits), (2) the concurrent mode learns (held-out accuracy), (3) files and round trips.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


@pytest.fixture(scope="module")
def amd():
    import opal_amd
    opal_amd.lib()
    return opal_amd


def _patterns(amd, golden_dir, fname, k):
    with open(os.path.join(golden_dir, fname)) as f:
        return amd.parse_patterns(f.readlines(), k)


def _margins(oracle, seqs, pos, w, K, bits):
    """Per fragment: the gap between the two largest (clipped) scores against table w, i.e. how decided the prediction is."""
    _, sc, _ = oracle.predict(seqs, 64, pos, True, w, K, bits)
    if K == 1:
        return np.full(len(seqs), np.inf)
    top = np.partition(sc.astype(np.float64), K - 2, axis=1)[:, K - 2:]
    return top[:, 1] - top[:, 0]


def _rows_of(oracle, seq, pos, K, bits):
    """The table slots fragment `seq` touches (its k-mer rows; not the constant row)."""
    mask = (1 << bits) - 1
    h = np.unique(oracle.feature_hashes(seq, 64, pos, True, hash_all=False).astype(np.uint64) & np.uint64(mask))
    return np.unique(((h[:, None] * np.uint64(K) + np.arange(K, dtype=np.uint64)[None, :]) & np.uint64(mask)).reshape(-1))


# (2100, 17) and (2101, 17): more classes than one register chunk holds (2 chunks of 2048) on tables small enough that rows of
# one example overlap across chunks: the rule is chunk-wise there (oracle/opal_oracle.c orc_train restates exactly that).
# Odd K (7, 355, 1565, 2101): rows start at any slot, their 16-byte vectors are dword aligned, the last one is partial, and on
# these small tables many rows pass the table's end (the slot-by-slot path).
@pytest.mark.parametrize("K,bits,l1,l2", [(10, 12, 0.0, 0.0), (356, 16, 0.0, 0.0), (7, 14, 1e-4, 1e-3), (1564, 18, 0.0, 0.0),
                                          (355, 20, 0.0, 0.0), (1565, 16, 0.0, 0.0), (2100, 17, 0.0, 0.0), (2101, 17, 1e-4, 0.0)])
def test_sequential_training_matches_oracle(amd, oracle, golden_dir, K, bits, l1, l2):
    rng = np.random.default_rng(K + bits)
    pos = _patterns(amd, golden_dir, "patterns_k64_t16_m1_s7.txt", 64)
    seqs = [bytes(rng.choice(ACGT, size=200)) for _ in range(24)]
    seqs[2] = b"A" * 200                       # every window identical: the same row 137 times in a row
    seqs[3] = b"ACAC" * 50                     # two alternating rows
    seqs[4] = seqs[4].lower()                  # raw path
    seqs[5] = seqs[5][:30]                     # no k-mer: only the constant feature
    seqs += seqs[:6]                           # repeats: weights already touched
    labels = rng.integers(1, K + 1, size=len(seqs)).astype(np.int32)
    eng = amd.Engine(0)
    eng.set_patterns(pos, 64)
    eng.train_begin(K, bits)
    batch = eng.upload(amd.PackedReads.from_sequences(seqs))
    w = np.zeros(1 << bits, dtype=np.float32)
    g2 = np.zeros(1 << bits, dtype=np.float32)
    for p in range(2):                         # two passes, like --num-passes 2
        pred = eng.train_batch(batch, labels, True, eta=0.5, l1=l1, l2=l2, sequential=True, want_pred=True)
        # the oracle one example at a time, so that the margin of every progressive prediction is known: the two may
        # differ only where the oracle's two best scores are closer than 1e-5 (GPU expf / v_rsq_f32 against libm)
        opred, margin = np.zeros(len(seqs), dtype=np.int32), np.zeros(len(seqs))
        for i in range(len(seqs)):
            margin[i] = _margins(oracle, seqs[i:i + 1], pos, w, K, bits)[0]
            opred[i] = oracle.train(seqs[i:i + 1], labels[i:i + 1], 64, pos, True, w, g2, K, bits, eta=0.5, l1=l1, l2=l2)[0]
        gw = eng.get_weights(to_host=True)
        assert np.allclose(gw, w, rtol=2e-4, atol=1e-6), (p, np.abs(gw - w).max())
        differ = pred.cpu().numpy() != opred
        assert not (differ & (margin >= 1e-5)).any(), (p, np.flatnonzero(differ).tolist(), margin[differ].tolist())
    assert np.count_nonzero(w) > 0
    # the trained table is immediately usable by predict (mirror padding refreshed)
    cls, sc = eng.predict(batch, True, want_scores=True)
    ocls, osc, _ = oracle.predict(seqs, 64, pos, True, gw, K, bits)
    assert np.array_equal(sc.cpu().numpy(), osc) and np.array_equal(cls.cpu().numpy(), ocls)
    eng.close()


@pytest.mark.parametrize("K,bits,l1,l2,group", [(10, 12, 0.0, 0.0, 7), (356, 16, 0.0, 0.0, 16), (7, 14, 1e-4, 1e-3, 5),
                                                (1564, 18, 0.0, 0.0, 64), (355, 20, 0.0, 0.0, 11), (1565, 16, 0.0, 0.0, 9),
                                                (2100, 17, 0.0, 0.0, 30), (2101, 17, 0.0, 0.0, 13)])
def test_deterministic_training_matches_oracle_and_itself(amd, oracle, golden_dir, K, bits, l1, l2, group):
    """opal_train_batch mode 2 (deterministic groups, opal_dtrain.hip): (1) against the CPU statement of the same rule
    (oracle.train_groups) at 2e-4 (GPU expf / v_rsq_f32 against libm), over several groups, a short last group, repeated
    and overlapping rows, text fragments, a fragment without k-mers; (2) two GPU runs give bit-identical tables and
    predictions, although partitions and examples are handed to waves in whatever order the hardware takes them."""
    rng = np.random.default_rng(K + bits)
    pos = _patterns(amd, golden_dir, "patterns_k64_t16_m1_s7.txt", 64)
    seqs = [bytes(rng.choice(ACGT, size=200)) for _ in range(40)]
    seqs[2] = b"A" * 200                       # every window identical: the same row 137 times in a row
    seqs[3] = b"ACAC" * 50                     # two alternating rows
    seqs[4] = seqs[4].lower()                  # raw path
    seqs[5] = seqs[5][:30]                     # no k-mer: only the constant feature
    seqs[6] = bytes(rng.choice(ACGT, size=333))
    seqs += seqs[:9]                           # repeats: weights already touched, the same rows again inside one group
    labels = rng.integers(1, K + 1, size=len(seqs)).astype(np.int32)
    pshift = oracle.train_pshift(K, bits)
    runs = []
    for rep in range(2):
        eng = amd.Engine(0)
        eng.set_patterns(pos, 64)
        eng.set_tuning(train_group=group)
        eng.train_begin(K, bits)
        batch = eng.upload(amd.PackedReads.from_sequences(seqs))
        w = np.zeros(1 << bits, dtype=np.float32)
        g2 = np.zeros(1 << bits, dtype=np.float32)
        got = []
        for p in range(2):                     # two passes
            pred = eng.train_batch(batch, labels, True, eta=0.5, l1=l1, l2=l2, sequential=2, want_pred=True)
            gw = eng.get_weights(to_host=True)
            got.append((gw, pred.cpu().numpy().copy()))
            if rep == 0:
                # the oracle a group at a time (the same thing as one call: groups are consecutive examples), so that the
                # margin of every progressive prediction against the table its group started from is known
                opred, margin = np.zeros(len(seqs), dtype=np.int32), np.zeros(len(seqs))
                for a in range(0, len(seqs), group):
                    margin[a:a + group] = _margins(oracle, seqs[a:a + group], pos, w, K, bits)
                    opred[a:a + group] = oracle.train_groups(seqs[a:a + group], labels[a:a + group], 64, pos, True, w, g2, K, bits,
                                                             group, pshift, eta=0.5, l1=l1, l2=l2)
                # 2e-4 everywhere but on the rows of the all-A fragment (seqs[2] and its repeat): it updates the same four rows
                # 137 times each per pass, their scores sit on the clip boundary, and expf / v_rsq_f32 against libm decide
                # on which side (seen: 4 slots at 5e-4).  WHICH slots may differ is stated, not how many:
                off_tol = np.flatnonzero(np.abs(gw - w) > 2e-4 * np.abs(w) + 1e-6)
                assert np.isin(off_tol, _rows_of(oracle, seqs[2], pos, K, bits)).all(), (p, off_tol.tolist())
                assert np.allclose(gw, w, rtol=2e-3, atol=1e-6), (p, np.abs(gw - w).max())
                differ = got[-1][1] != opred
                assert not (differ & (margin >= 1e-5)).any(), (p, np.flatnonzero(differ).tolist(), margin[differ].tolist())
        runs.append(got)
        if rep == 0:
            assert np.count_nonzero(w) > 0
            cls, sc = eng.predict(batch, True, want_scores=True)          # mirror padding refreshed: usable at once
            ocls, osc, _ = oracle.predict(seqs, 64, pos, True, gw, K, bits)
            assert np.array_equal(sc.cpu().numpy(), osc) and np.array_equal(cls.cpu().numpy(), ocls)
        eng.close()
    for (wa, pa), (wb, pb) in zip(*runs):
        assert np.array_equal(wa, wb) and np.array_equal(pa, pb)


def test_concurrent_training_with_regularisation_keeps_the_shared_row(amd, golden_dir):
    """--lambda1/--lambda2 in the concurrent mode: the constant row, which every example updates, goes through a
    compare-and-swap loop (decay and soft threshold are not additive), so no example's step on it is lost: with a tiny
    regularisation its weights end up where the unregularised run's float atomics put them.  (With a plain
    read-modify-write most concurrent steps on that row vanished.)  The sequential run is there for the signs only: a
    concurrent run overshoots it on this row, because thousands of examples read the bias before any has moved it."""
    rng = np.random.default_rng(3)
    K, bits = 8, 22
    pos = _patterns(amd, golden_dir, "patterns_k64_t16_m1_s7.txt", 64)
    seqs = [bytes(rng.choice(ACGT, size=200)) for _ in range(6000)]       # unrelated fragments: only the constant row is shared
    labels = np.ones(len(seqs), dtype=np.int32)                          # one class always right: its bias must climb
    out = {}
    for name, sequential, l1, l2 in (("seq", True, 1e-7, 1e-6), ("conc", False, 0.0, 0.0), ("conc_reg", False, 1e-7, 1e-6)):
        eng = amd.Engine(0)
        eng.set_patterns(pos, 64)
        eng.train_begin(K, bits)
        batch = eng.upload(amd.PackedReads.from_sequences(seqs))
        eng.train_batch(batch, labels, True, eta=0.5, l1=l1, l2=l2, sequential=sequential)
        w = eng.get_weights(to_host=True)
        crow = (11650396 & ((1 << bits) - 1)) * K & ((1 << bits) - 1)
        out[name] = w[crow:crow + K].copy()
        eng.close()
    for v in out.values():
        assert v[0] > 0.05 and np.all(v[1:] < -0.05)
    # both runs are Hogwild (how stale the bias is that an example reads depends on timing: 0.150 against 0.167 has been seen);
    # a lost-update bug leaves the regularised row an order of magnitude short
    assert np.allclose(out["conc_reg"], out["conc"], rtol=0.35), (out["conc_reg"], out["conc"])


def _make_genomes(rng, n_genomes, glen, divergence):
    """n_genomes related genomes: mutated copies of a common ancestor (shared k-mers make the task non-trivial)."""
    anc = rng.integers(0, 4, size=glen, dtype=np.uint8)
    out = []
    for _ in range(n_genomes):
        g = anc.copy()
        m = rng.random(glen) < divergence
        g[m] = (g[m] + rng.integers(1, 4, size=int(m.sum()))) % 4
        out.append(bytes(ACGT[g]))
    return out


def test_concurrent_training_learns_and_model_roundtrip(amd, golden_dir, tmp_path):
    from opal_amd import model as model_io
    rng = np.random.default_rng(5)
    K, bits, L = 12, 26, 200
    genomes = _make_genomes(rng, K, 30_000, 0.08)
    pos = _patterns(amd, golden_dir, "patterns_k64_t16_m1_s7.txt", 64)

    def draw(n):
        g = rng.integers(0, K, size=n)
        p = rng.integers(0, 30_000 - L, size=n)
        return [genomes[a][b:b + L] for a, b in zip(g, p)], (g + 1).astype(np.int32)

    eng = amd.Engine(0)
    eng.set_patterns(pos, 64)
    eng.train_begin(K, bits)
    tr_seqs, tr_lab = draw(20_000)
    batch = eng.upload(amd.PackedReads.from_sequences(tr_seqs))
    errs = []
    for _ in range(2):
        pred = eng.train_batch(batch, tr_lab, True, want_pred=True)
        errs.append(float((pred.cpu().numpy() != tr_lab).mean()))
    assert errs[1] < errs[0]
    te_seqs, te_lab = draw(4_000)
    tb = eng.upload(amd.PackedReads.from_sequences(te_seqs))
    cls, _ = eng.predict(tb, True)
    acc = float((cls.cpu().numpy() == te_lab).mean())
    assert acc > 0.9, acc
    # container round trip (sparse payload) gives identical predictions
    path = str(tmp_path / "vw-model_final.model")
    nnz = model_io.save_model(path, eng.get_weights(), K, bits)
    assert 0 < nnz < (1 << bits) // 2 and not model_io.read_header(path)["dense"]
    e2 = amd.Engine(0)
    e2.set_patterns(pos, 64)
    hdr = model_io.load_model(path, e2)
    assert (hdr["K"], hdr["bits"]) == (K, bits)
    cls2, _ = e2.predict(tb, True)
    assert np.array_equal(cls2.cpu().numpy(), cls.cpu().numpy())
    # dense payload
    w = (rng.standard_normal(1 << 12) * 0.1).astype(np.float32)
    p2 = str(tmp_path / "dense_final.model")
    model_io.save_model(p2, w, 5, 12, clip=False)
    h2 = model_io.load_model(p2, e2)
    assert h2["dense"] and not h2["clip"]
    assert np.array_equal(e2.get_weights(to_host=True), w)
    eng.close()
    e2.close()


def test_cli_simulate_pipeline(amd, tmp_path, capsys):
    """opal.py simulate: frag -> train -> predict -> eval with the reference's directory/file contract."""
    from opal_amd import opal
    rng = np.random.default_rng(9)
    K = 6
    genomes = _make_genomes(rng, K, 20_000, 0.10)
    taxids = ["562", "1280", "1313", "4932", "9606", "562x"]
    for d in ("train", "test"):
        os.makedirs(tmp_path / d)
        with open(tmp_path / d / "A1.fasta", "w") as f:
            for i, g in enumerate(genomes):
                f.write(">gi|%d| genome %d\n" % (i, i))
                f.write("\n".join(g[j:j + 70].decode() for j in range(0, len(g), 70)) + "\n")
        with open(tmp_path / d / "A1.taxid", "w") as f:
            f.write("".join(t + "\n" for t in taxids))
    out = tmp_path / "out"
    np.random.seed(1)
    opal.main(["simulate", str(tmp_path / "test"), str(tmp_path / "train"), str(out), "-r", "-l", "200", "-c", "2",
               "--num-hash", "2", "--num-batches", "2", "--bits", "22"])
    for rel in ("1frag/test.fragments.fasta", "1frag/test.fragments.gi2taxid", "1frag/test.fragments.taxid",
                "2model/patterns.txt", "2model/vw-dico.txt", "2model/vw-model_final.model", "2model/vw-model_vwps.log",
                "3predict/test.fragments-db.preds.vw", "3predict/test.fragments-db.preds.taxid",
                "3predict/test.fragments-db_vwps.log"):
        assert os.path.isfile(out / rel), rel
    assert not list((out / "2model").glob("train.batch-*"))         # transient batch files removed
    assert open(out / "2model" / "patterns.txt").readline() == "3 16\n"
    assert open(out / "2model" / "vw-dico.txt").read().count("\n") == K
    text = capsys.readouterr().out
    micro = float([l for l in text.splitlines() if l.startswith("micro = ")][-1].split("=")[1])
    assert micro > 0.9, text[-2000:]
    ref = open(out / "1frag" / "test.fragments.taxid").read().splitlines()
    pred = open(out / "3predict" / "test.fragments-db.preds.taxid").read().splitlines()
    assert len(ref) == len(pred) == 6 * 200
    # eval sub-command on the same files prints the same numbers
    opal.main(["eval", str(out / "1frag" / "test.fragments.taxid"), str(out / "3predict" / "test.fragments-db.preds.taxid")])
    assert "micro = %.4f" % micro in capsys.readouterr().out


def test_cli_train_is_reproducible_by_default(amd, tmp_path):
    """The reference's learner is one sequential vw process: the same inputs give the same model.  `opal.py train` keeps that
    with its default schedule (deterministic groups): two runs write byte-identical vw-model_final.model files and log the
    schedule; --train-mode concurrent (Hogwild) still trains a usable model."""
    from opal_amd import opal
    rng = np.random.default_rng(10)
    genomes = _make_genomes(rng, 5, 30_000, 0.08)
    os.makedirs(tmp_path / "train")
    with open(tmp_path / "train" / "A1.fasta", "w") as f:
        for i, g in enumerate(genomes):
            f.write(">g%d\n%s\n" % (i, g.decode()))
    with open(tmp_path / "train" / "A1.taxid", "w") as f:
        f.write("".join("%d\n" % (100 + i) for i in range(5)))
    blobs = []
    for run in ("a", "b"):
        np.random.seed(4)                                   # same patterns.txt in both runs (ldpc.py uses numpy's global stream)
        opal.main(["train", str(tmp_path / "train"), str(tmp_path / run), "-r", "-l", "200", "-c", "3", "--num-hash", "1",
                   "--num-batches", "2", "--bits", "24"])
        blobs.append(open(tmp_path / run / "vw-model_final.model", "rb").read())
        assert "schedule=deterministic" in open(tmp_path / run / "vw-model_vwps.log").readline()
    assert blobs[0] == blobs[1] and len(blobs[0]) > 10_000
    np.random.seed(4)
    opal.main(["train", str(tmp_path / "train"), str(tmp_path / "c"), "-r", "-l", "200", "-c", "3", "--num-hash", "1",
               "--num-batches", "2", "--bits", "24", "--train-mode", "concurrent"])
    assert "schedule=concurrent" in open(tmp_path / "c" / "vw-model_vwps.log").readline()
    assert os.path.getsize(tmp_path / "c" / "vw-model_final.model") > 10_000
    # --save-resume / --resume: two batches in one run == one batch, saved with the learner's state, then one more batch in a
    # second process-worth of state (fresh engine, model and accumulators from the file, seeds continued): identical files
    np.random.seed(4)
    opal.main(["train", str(tmp_path / "train"), str(tmp_path / "d"), "-r", "-l", "200", "-c", "3", "--num-hash", "1",
               "--num-batches", "2", "--bits", "24", "--save-resume"])
    np.random.seed(4)
    opal.main(["train", str(tmp_path / "train"), str(tmp_path / "e"), "-r", "-l", "200", "-c", "3", "--num-hash", "1",
               "--num-batches", "1", "--bits", "24", "--save-resume"])
    opal.main(["train", str(tmp_path / "train"), str(tmp_path / "e"), "-r", "-l", "200", "-c", "3", "--num-hash", "1",
               "--num-batches", "1", "--save-resume", "--resume"])
    d, e = (open(tmp_path / x / "vw-model_final.model", "rb").read() for x in ("d", "e"))
    assert d == e and len(d) > len(blobs[0])
    # the weights section of the resumable file is the plain file's
    from opal_amd import model as model_io
    assert model_io.read_header(str(tmp_path / "d" / "vw-model_final.model"))["batches"] == 2
    with pytest.raises(RuntimeError, match="no resume state"):
        opal.main(["train", str(tmp_path / "train"), str(tmp_path / "a"), "-r", "-l", "200", "-c", "3", "--num-hash", "1",
                   "--num-batches", "1", "--resume"])


def test_cli_train_multipass_holdout_and_early_stop(amd, tmp_path):
    """--num-passes > 1 (reference opal.py:316-320: vw --cache_file --passes N on the whole stream): the stream of all
    batches is replayed, every 10th example is held out, the log carries one holdout line per pass and training stops
    after 3 passes without improvement; the saved model still classifies the training genomes."""
    import argparse
    from opal_amd import opal, model as model_io
    rng = np.random.default_rng(19)
    K = 5
    genomes = _make_genomes(rng, K, 20_000, 0.30)
    os.makedirs(tmp_path / "train")
    with open(tmp_path / "train" / "A1.fasta", "w") as f:
        for i, g in enumerate(genomes):
            f.write(">g%d\n%s\n" % (i, g.decode()))
    with open(tmp_path / "train" / "A1.taxid", "w") as f:
        f.write("".join("t%d\n" % i for i in range(K)))
    np.random.seed(3)
    args = argparse.Namespace(frag_length=200, coverage=3.0, kmer=64, row_weight=16, hierarchical_weight=-1, num_hash=1,
                              num_batches=2, num_passes=8, bits=22, lambda1=0.0, lambda2=0.0, reverse_complement=True)
    opal.train(str(tmp_path / "train"), str(tmp_path / "model"), args)
    log = open(tmp_path / "model" / "vw-model_vwps.log").read().splitlines()
    first = [l for l in log if " pass 0: " in l and l.startswith("batch")]
    assert len(first) == 2
    n_trained = sum(int(l.split(":")[1].split()[0]) for l in first)
    assert n_trained == 2 * 1500 - (2 * 1500) // 10                   # every 10th example of the stream is held out
    hold = [l for l in log if "holdout error" in l]
    assert 2 <= len(hold) <= 8 and hold[0].startswith("pass 0:")
    errs = [float(l.rsplit(" ", 1)[1]) for l in hold]
    if len(hold) < 8:                                                  # stopped early: 3 passes in a row without a new best
        assert any("no improvement in 3 passes" in l for l in log)
        assert min(errs[-3:]) >= min(errs[:-3])
    eng = amd.Engine(0)
    with open(tmp_path / "model" / "patterns.txt") as f:
        pos = amd.parse_patterns(f.readlines(), 64)
    eng.set_patterns(pos, 64)
    model_io.load_model(str(tmp_path / "model" / "vw-model_final.model"), eng)
    frags = [g[a:a + 200] for g in genomes for a in range(0, 19_000, 1000)]
    cls, _ = eng.predict(eng.upload(amd.PackedReads.from_sequences(frags)), True)
    truth = np.repeat(np.arange(1, K + 1), 19)
    assert (cls.cpu().numpy() == truth).mean() > 0.95
    eng.close()


def test_gpu_eval_counts_match_reference_output(amd, golden_dir, tmp_path, capsys):
    """opal_eval_counts (through evaluate_predictions_gpu) prints what the reference's evaluate_predictions printed
    (tests/golden/opal_helpers.json, reference opal.py:111-138), and refuses files of unequal length like it."""
    import json
    from opal_amd import opal
    with open(os.path.join(golden_dir, "opal_helpers.json")) as f:
        cases = json.load(f)["evaluate_predictions"]
    eng = amd.Engine(0)
    rf, pf = tmp_path / "ref.taxid", tmp_path / "pred.taxid"
    capsys.readouterr()
    for case in cases:
        rf.write_text("".join(x + "\n" for x in case["ref"]))
        pf.write_text("".join(x + "\n" for x in case["pred"]))
        if case["error"]:
            with pytest.raises(ValueError):
                opal.evaluate_predictions_gpu(str(rf), str(pf), eng)
            continue
        opal.evaluate_predictions_gpu(str(rf), str(pf), eng)
        assert capsys.readouterr().out == case["stdout"]
    eng.close()


def test_training_stream_vw_text_matches_reference(amd, golden_dir):
    """End to end pin of the train data path: the examples of draw_training_batch, turned into VW text lines with the
    GPU-gathered feature names, are byte for byte what the reference pipes to vw for seeds 421, 422, ... (sha256 of the
    shuffled training_list, tests/golden/train_stream.json; reference opal.py:329-367)."""
    import hashlib
    import io
    import json
    import tempfile
    from opal_amd import fasta2skm, opal
    from opal_amd.fasta_functions import fasta_reader
    with open(os.path.join(golden_dir, "train_stream.json")) as f:
        cases = json.load(f)
    eng = amd.Engine(0)
    for case in cases:
        with open(os.path.join(golden_dir, case["pattern"])) as f:
            eng.set_patterns(amd.parse_patterns(f.readlines(), case["kmer"]), case["kmer"])
        genomes = [s for _, s in fasta_reader(io.StringIO(case["fasta"]))]
        taxid_lines = case["taxid"].splitlines()
        genome_taxid = [taxid_lines[g] if g < len(taxid_lines) else "" for g in range(len(genomes))]
        L = case["frag_length"]
        rng = amd.Rng()
        with tempfile.TemporaryDirectory() as d:
            for b in case["batches"]:
                vwid, pos, gen = opal.draw_training_batch(rng, b["seed"], [len(x) for x in genomes], genome_taxid, L,
                                                          case["coverage"], os.path.join(d, "vw-dico.txt"))
                frags = [genomes[g][p:p + L].encode("latin-1") for p, g in zip(pos.tolist(), gen.tolist())]
                text = b"".join(fasta2skm.vw_text_lines(eng, frags, case["reverse"], vwid.tolist()))
                assert len(text) == b["vw_text_bytes"]
                assert hashlib.sha256(text).hexdigest() == b["vw_text_sha256"]
    eng.close()


def test_gpu_eval_counts_match_host(amd, oracle, tmp_path, capsys):
    from opal_amd import opal
    rng = np.random.default_rng(1)
    n, C = 50_000, 37
    ref = rng.integers(0, C, size=n)
    pred = np.where(rng.random(n) < 0.8, ref, rng.integers(0, C + 3, size=n))
    names = ["tax%d" % i for i in range(C + 3)]
    rf, pf = tmp_path / "ref.taxid", tmp_path / "pred.taxid"
    rf.write_text("".join(names[i] + "\n" for i in ref))
    pf.write_text("".join(names[i] + "\n" for i in pred))
    eng = amd.Engine(0)
    got = opal.evaluate_predictions_gpu(str(rf), str(pf), eng)
    host = opal.evaluate_predictions(str(rf), str(pf))
    want = oracle.evaluate([names[i] for i in ref], [names[i] for i in pred])
    assert np.allclose(got, want, atol=1e-12) and np.allclose(host, want, atol=1e-12)
    # many classes: the global-atomic path
    C2 = 20_000
    r2 = rng.integers(0, C2, size=n).astype(np.int32)
    p2 = np.where(rng.random(n) < 0.5, r2, 0).astype(np.int32)
    cor, tot = eng.eval_counts(p2, r2, C2)
    assert np.array_equal(tot, np.bincount(r2, minlength=C2).astype(np.uint64))
    assert np.array_equal(cor, np.bincount(r2[p2 == r2], minlength=C2).astype(np.uint64))
    eng.close()


def test_vw_readable_model_import(amd, tmp_path):
    """Best-effort text importer (layout from VW 8.x documentation; unverified against a real vw)."""
    from opal_amd import model as model_io
    p = tmp_path / "m.txt"
    p.write_text("Version 8.6.1\nId \nMin label:-1\nMax label:1\nbits:12\nlda:0\n0 ngram:\n0 skip:\n"
                 "options: --oaa 3 --random_seed 420\nChecksum: 1\n:0\n5:0.25\n17:-1.5 0.3 1\n4095:2\n")
    m = model_io.import_vw_readable_model(str(p))
    assert (m["K"], m["bits"]) == (3, 12) and m["idx"].tolist() == [5, 17, 4095] and m["val"].tolist() == [0.25, -1.5, 2.0]
    eng = amd.Engine(0)
    eng.load_weights_sparse(m["idx"], m["val"], m["K"], m["bits"])
    w = eng.get_weights(to_host=True)
    assert w[5] == 0.25 and w[17] == -1.5 and w[4095] == 2.0 and np.count_nonzero(w) == 3
    eng.close()


def test_saturated_classes_skip_updates_but_keep_the_rule(amd, oracle, golden_dir):
    """A model whose negatives already sit at -1 (err == 0 after the clip): the kernel skips those classes' rows, the
    oracle's sequential rule adds zeros; both must leave the same table."""
    rng = np.random.default_rng(12)
    with open(os.path.join(golden_dir, "patterns_k64_t16_m1_s7.txt")) as f:
        pos = amd.parse_patterns(f.readlines(), 64)
    K, bits = 40, 16
    seqs = [bytes(rng.choice(ACGT, size=200)) for _ in range(20)]
    labels = rng.integers(1, K + 1, size=len(seqs)).astype(np.int32)
    w0 = np.full(1 << bits, -0.01, dtype=np.float32)      # 549 features x -0.01 = -5.49 -> every class clips to -1
    eng = amd.Engine(0)
    eng.set_patterns(pos, 64)
    eng.load_weights(w0, K, bits)
    eng.train_begin(K, bits, resume=True)
    batch = eng.upload(amd.PackedReads.from_sequences(seqs))
    w, g2 = w0.copy(), np.zeros(1 << bits, dtype=np.float32)
    eng.train_batch(batch, labels, True, sequential=True)
    oracle.train(seqs, labels, 64, pos, True, w, g2, K, bits)
    gw = eng.get_weights(to_host=True)
    assert np.allclose(gw, w, rtol=2e-4, atol=1e-6)
    assert (gw != w0).sum() > 0 and (gw != w0).mean() < 0.5      # only the true classes' slots moved
    eng.close()


def _predict_rank(rank, world, port, model_dir, test_dir, out_dir):
    import argparse
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from opal_amd import opal
    # pieces of ~10 kB: the 1001 reads make about 18 pieces, dealt to the ranks in turn
    opal.predict(model_dir, test_dir, out_dir, argparse.Namespace(kmer=64, reverse_complement=True), chunk_bytes=10_000)


def test_cli_predict_two_ranks_equals_one(amd, tmp_path):
    """`opal.py predict` launched as two ranks (here both on the one GPU of the box): each scores its own pieces of the file,
    rank 0 writes the files; the result equals the single-process run byte for byte."""
    import argparse
    import socket
    import torch.multiprocessing as mp
    from opal_amd import opal, model as model_io
    rng = np.random.default_rng(23)
    K, bits = 9, 18
    os.makedirs(tmp_path / "model")
    os.makedirs(tmp_path / "test")
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "golden", "patterns_k64_t16_m1_s7.txt")) as f:
        (tmp_path / "model" / "patterns.txt").write_text(f.read())
    (tmp_path / "model" / "vw-dico.txt").write_text("".join("tax%d\t%d\n" % (i, i) for i in range(1, K + 1)))
    w = (rng.standard_normal(1 << bits) * 0.05).astype(np.float32)
    model_io.save_model(str(tmp_path / "model" / "vw-model_final.model"), w, K, bits)
    with open(tmp_path / "test" / "reads.fasta", "w") as f:
        for i in range(1001):
            f.write(">r%d\n%s\n" % (i, bytes(ACGT[rng.integers(0, 4, size=int(rng.integers(40, 300)))]).decode()))
    args = argparse.Namespace(kmer=64, reverse_complement=True)
    opal.predict(str(tmp_path / "model"), str(tmp_path / "test"), str(tmp_path / "one"), args)
    # the same file streamed in header-aligned pieces of ~10 kB (what happens to files beyond a few GiB)
    opal.predict(str(tmp_path / "model"), str(tmp_path / "test"), str(tmp_path / "pieces"), args, chunk_bytes=10_000)
    assert (tmp_path / "pieces" / "test.fragments-db.preds.vw").read_bytes() == (tmp_path / "one" / "test.fragments-db.preds.vw").read_bytes()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_predict_rank, args=(2, port, str(tmp_path / "model"), str(tmp_path / "test"), str(tmp_path / "two")), nprocs=2, join=True)
    for name in ("test.fragments-db.preds.vw", "test.fragments-db.preds.taxid"):
        a, b = (tmp_path / "one" / name).read_bytes(), (tmp_path / "two" / name).read_bytes()
        assert a == b and a.count(b"\n") == 1001


def test_resume_state_continues_the_rule(amd, golden_dir, tmp_path):
    """--save_resume (reference opal.py:310): train A then B in one run == train A, save the model WITH its accumulators
    (version-2 container), load it into a fresh context, train_begin(resume=True), train B: bit-identical weights and
    accumulators in the deterministic schedule.  A version-1 file (weights only) refuses to resume; resuming from weights
    alone (accumulators restarted at zero) gives a different table, which is what the state is for."""
    from opal_amd import model as model_io
    rng = np.random.default_rng(21)
    K, bits = 37, 18
    pos = _patterns(amd, golden_dir, "patterns_k64_t16_m1_s7.txt", 64)
    A = [bytes(rng.choice(ACGT, size=200)) for _ in range(300)]
    B = [bytes(rng.choice(ACGT, size=200)) for _ in range(200)] + A[:50]
    la = rng.integers(1, K + 1, size=len(A)).astype(np.int32)
    lb = rng.integers(1, K + 1, size=len(B)).astype(np.int32)

    def fresh():
        e = amd.Engine(0)
        e.set_patterns(pos, 64)
        e.set_tuning(train_group=64)
        return e

    one = fresh()
    one.train_begin(K, bits)
    one.train_batch(one.upload(amd.PackedReads.from_sequences(A)), la, True, sequential=2)
    path, path_v1 = str(tmp_path / "resumable.model"), str(tmp_path / "plain.model")
    model_io.save_model(path, one.get_weights(), K, bits, g=one.get_accumulators(), batches=1)
    model_io.save_model(path_v1, one.get_weights(), K, bits)
    assert model_io.read_header(path)["resumable"] and model_io.read_header(path)["batches"] == 1
    assert not model_io.read_header(path_v1)["resumable"]
    assert os.path.getsize(path) > os.path.getsize(path_v1)
    one.train_batch(one.upload(amd.PackedReads.from_sequences(B)), lb, True, sequential=2)
    w_one, g_one = one.get_weights(to_host=True), one.get_accumulators().cpu().numpy()
    one.close()

    two = fresh()
    model_io.load_model(path, two, resume=True)
    two.train_begin(K, bits, resume=True)
    two.train_batch(two.upload(amd.PackedReads.from_sequences(B)), lb, True, sequential=2)
    assert np.array_equal(two.get_weights(to_host=True), w_one)
    assert np.array_equal(two.get_accumulators().cpu().numpy(), g_one)
    two.close()

    three = fresh()
    with pytest.raises(RuntimeError, match="no resume state"):
        model_io.load_model(path_v1, three, resume=True)
    model_io.load_model(path_v1, three)                      # weights only: the accumulators restart at zero
    three.train_begin(K, bits, resume=True)
    three.train_batch(three.upload(amd.PackedReads.from_sequences(B)), lb, True, sequential=2)
    assert not np.array_equal(three.get_weights(to_host=True), w_one)
    three.close()


def test_training_fuzz(amd, oracle, golden_dir):
    """Seeded fuzz of opal_train_batch mode 2 against oracle.train_groups (and, a third of the trials, mode 1 against oracle.train) over what the kernels specialise on: the class count
    (1 .. 40, and around every multiple of 256 = one pass of 64 lanes x 4 classes, up to 2 chunks of 2048), the table size
    (rows overlap and wrap on the small ones), the group size (1 .. 50: short last groups, groups of one), numHash 1 and 4,
    one or both strands, regularisation on and off, fragments of 0 .. 400 bases, some as text.  Two passes: the second starts
    from live weights, where the saturated classes are skipped (phase A's live-vector bits, phase B's fetches).
    OPAL_FUZZ_SEED / OPAL_FUZZ_TRIALS: longer campaigns with other seeds (run by hand on the GPU box)."""
    rng = np.random.default_rng(int(os.environ.get("OPAL_FUZZ_SEED", "9107")))
    pats = [_patterns(amd, golden_dir, f, 64) for f in ("patterns_k64_t16_m1_s7.txt", "patterns_k64_t16_m4_s7.txt")]
    near = [256, 512, 1024, 2048]
    worst = 0.0
    for trial in range(int(os.environ.get("OPAL_FUZZ_TRIALS", "14"))):
        kind = trial % 3
        K = int(rng.integers(1, 41)) if kind == 0 else int(near[rng.integers(0, 4)] + rng.integers(-3, 5)) if kind == 1 else int(rng.integers(41, 2300))
        bits = int(rng.integers(max(10, int(np.ceil(np.log2(2 * K)))), 19))
        group = int(rng.integers(1, 51))
        reverse = bool(rng.integers(0, 2))
        pos = pats[int(rng.integers(0, 2))]
        l1, l2 = ((1e-4, 1e-3) if rng.random() < 0.25 else (0.0, 0.0))
        mode = 2 if rng.random() < 0.65 else 1        # 1: the one-wave sequential schedule (train_kernel) against oracle.train
        seqs = []
        for _ in range(int(rng.integers(1, 36))):
            L = int(rng.choice([0, 30, 63, 64, 65, 100, 200, 200, 200, 257, 400]))
            s = bytes(rng.choice(ACGT, size=L))
            seqs.append(s.lower() if rng.random() < 0.15 else s)
        labels = rng.integers(1, K + 1, size=len(seqs)).astype(np.int32)
        eng = amd.Engine(0)
        eng.set_patterns(pos, 64)
        eng.set_tuning(train_group=group)
        eng.train_begin(K, bits)
        batch = eng.upload(amd.PackedReads.from_sequences(seqs))
        w = np.zeros(1 << bits, dtype=np.float32)
        g2 = np.zeros(1 << bits, dtype=np.float32)
        pshift = oracle.train_pshift(K, bits)
        what = dict(trial=trial, mode=mode, K=K, bits=bits, group=group, reverse=reverse, P=int(pos.shape[0]), l1=l1, n=len(seqs))
        for p in range(2):
            eng.train_batch(batch, labels, reverse, eta=0.5, l1=l1, l2=l2, sequential=mode)
            if mode == 2:
                oracle.train_groups(seqs, labels, 64, pos, reverse, w, g2, K, bits, group, pshift, eta=0.5, l1=l1, l2=l2)
            else:
                oracle.train(seqs, labels, 64, pos, reverse, w, g2, K, bits, eta=0.5, l1=l1, l2=l2)
            gw = eng.get_weights(to_host=True)
            gg = eng.get_accumulators().cpu().numpy()
            # GPU expf / v_rsq_f32 against libm, through as many steps as a slot takes: on the small tables here a slot belongs to
            # hundreds of overlapping rows and its weight is what is left of steps that cancel, so the deviation is bounded against
            # the table's scale, not per slot.  Seen over 1 500 trials x 2 passes (seed 424242): at most 2.2e-4 of max |w| (and of
            # max G), the worst on 2^10 .. 2^12-slot tables with regularisation; 2e-4 per slot on weights above 1e-3 otherwise.
            assert np.abs(gw - w).max() <= 5e-4 * np.abs(w).max() + 1e-7, (what, p, float(np.abs(gw - w).max()), float(np.abs(w).max()))
            assert np.abs(gg - g2).max() <= 5e-4 * np.abs(g2).max() + 1e-7, (what, p, float(np.abs(gg - g2).max()), float(np.abs(g2).max()))
            nz = np.abs(w) > 1e-3
            if nz.any():
                worst = max(worst, float((np.abs(gw - w)[nz] / np.abs(w)[nz]).max()))
        eng.close()
    print("trainer fuzz: worst relative deviation of a weight from the oracle's %.2e" % worst)
