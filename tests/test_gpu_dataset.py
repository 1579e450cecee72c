"""Training-sample assembly on the GPU (videometamaterials_amd.Dataset, vmm_fields_to_samples) against fixtures produced by the
reference's own Dataset (tests/golden/make_golden_dataset.py) and against the oracle: bit-exact."""
import os

import numpy as np
import pytest
import torch

import helpers
from test_dataset_oracle import load_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(helpers.DATASET_CASES))
def test_minibatch_assembly_matches_reference_dataset(gpu, name):
    import videometamaterials_amd as vm
    frame, N, num_frames, sel, per_frame, frames, fr, curves, gold = load_case(name)
    ds = vm.Dataset(frames, fr, curves, selected_channels=sel, num_frames=num_frames, per_frame_cond=per_frame, reference_frame=frame, device=gpu)
    assert len(ds) == N
    assert np.array_equal(ds.labels.numpy(), gold["labels"])
    if frame == "lagrangian":
        assert np.array_equal(ds.zero_u_2.numpy(), gold["zero_u_2"])
    for k in gold.files:
        if k.startswith("g_"):
            assert float(getattr(ds, k[2:])) == float(gold[k])
    x, lab = ds.batch(list(range(N)))
    want = np.stack([gold[f"sample{i}"] for i in range(N)])
    assert x.shape == want.shape and x.dtype == torch.float32
    assert np.array_equal(x.cpu().numpy(), want)
    assert np.array_equal(lab.cpu().numpy(), gold["labels"])
    # any order, repeats, one at a time
    idx = [N - 1, 0, 0, N // 2]
    x2, lab2 = ds.batch(torch.tensor(idx))
    assert np.array_equal(x2.cpu().numpy(), want[idx]) and np.array_equal(lab2.cpu().numpy(), gold["labels"][idx])
    xi, li = ds[1]
    assert np.array_equal(xi.cpu().numpy(), want[1]) and np.array_equal(li.cpu().numpy(), gold["labels"][1])
    with pytest.raises(IndexError):
        ds.batch([N])
    # negative indices count from the end (the reference indexes Python lists); device-resident index tensors take the no-sync path
    x3, _ = ds.batch([-1, -N])
    assert np.array_equal(x3.cpu().numpy(), want[[N - 1, 0]])
    x4, lab4 = ds.batch(torch.tensor([1, -1], device=gpu))
    assert np.array_equal(x4.cpu().numpy(), want[[1, N - 1]]) and np.array_equal(lab4.cpu().numpy(), gold["labels"][[1, N - 1]])
    with pytest.raises(IndexError):
        ds.batch([-N - 1])
    # device-resident indices out of range: no synchronisation inside batch() (the rows are clamped for the launch), the error is latched on
    # the device and raised by check_indices() at the caller's next synchronisation point, then cleared
    ds.check_indices()
    x5, _ = ds.batch(torch.tensor([0, N + 3], device=gpu))
    assert np.array_equal(x5.cpu().numpy(), want[[0, N - 1]])
    ds.batch(torch.tensor([1], device=gpu))  # a later good batch does not clear the latch
    with pytest.raises(IndexError):
        ds.check_indices()
    ds.check_indices()


def test_min_max_values_csv_like_the_reference(gpu, tmp_path):
    """<folder>/min_max_values.csv (vddp.py:1210-1246): the rows, their order and values of the reference's constructor."""
    import csv
    import videometamaterials_amd as vm
    for frame, names in (("lagrangian", ["min_u_1", "max_u_1", "min_u_2", "max_u_2", "max_s_mises", "min_s_22", "max_s_22", "max_strain_energy"]),
                         ("eulerian", ["max_s_mises", "min_s_22", "max_s_22", "max_strain_energy"])):
        frames, fr, curves = helpers.synth_dataset(5, frame, 4, 3, 8)
        ds = vm.Dataset(torch.from_numpy(frames), fr, curves, reference_frame=frame, selected_channels=[0, 1], num_frames=3, device=gpu)
        path = ds.write_min_max_values(str(tmp_path))
        rows = list(csv.reader(open(path)))
        assert [r[0] for r in rows] == names
        fr = np.asarray(fr, dtype=np.float64)
        cols = dict(zip(names, range(8))) if frame == "lagrangian" else dict(zip(names, range(4)))
        for k, v in rows:
            col = fr[:, cols[k]]
            assert float(v) == (col.min() if k.startswith("min") else col.max()), k


@pytest.mark.parametrize("P,frame", [(15, "lagrangian"), (96, "lagrangian"), (7, "eulerian")])
def test_minibatch_assembly_matches_oracle(gpu, P, frame):
    """Sizes without fixtures: a pixel count that is not a multiple of four (scalar path), and the full 96 x 96 frames."""
    import videometamaterials_amd as vm
    from oracle import dataset_oracle as do
    N, f = 6, 11
    frames, fr, curves = helpers.synth_dataset(90 + P, frame, N, f, P)
    frames, fr = torch.from_numpy(frames), torch.tensor(fr)
    sel = [0, 1, 3]
    ds = vm.Dataset(frames, fr, curves, selected_channels=sel, num_frames=11, per_frame_cond=True, reference_frame=frame, device=gpu)
    g = do.global_ranges(fr, frame)
    want = torch.stack([do.fields_to_sample(frames[i], fr[i], g, frame, sel, 11) for i in range(N)])
    x, _ = ds.batch(range(N))
    assert torch.equal(x.cpu(), want)


def test_dataset_rejections(gpu):
    import videometamaterials_amd as vm
    frames, fr, curves = helpers.synth_dataset(1, "lagrangian", 2, 3, 8)
    with pytest.raises(NotImplementedError):
        vm.Dataset(torch.from_numpy(frames), fr, curves, horizontal_flip=True, reference_frame="lagrangian", device=gpu)
    with pytest.raises(ValueError):
        vm.Dataset(torch.from_numpy(frames), fr, curves, reference_frame="eulerian", device=gpu)  # five fields given, four expected
    with pytest.raises(ValueError):
        vm.Dataset(torch.from_numpy(frames).float(), fr, curves, reference_frame="lagrangian", device=gpu)
