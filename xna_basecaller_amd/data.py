"""
Validation data of `bonito evaluate` (ub-bonito/bonito/data.py:104-163): a directory with chunks.npy (N, L) signal
chunks, references.npy (N, Lmax) integer-coded references (0 = padding) and reference_lengths.npy (N,), optionally an
indices.npy subsample and a validation/ sub-directory.  Only what evaluation needs: the reference's training-side
augmentation (spiking / stitching of XNA segments) is outside the MI355X path.
"""
import os

import numpy as np


def load_numpy_datasets(limit=None, directory=None):
    """(chunks, targets, lengths) of a ctc-data directory, subsampled by indices.npy when present, cut to `limit`."""
    chunks = np.load(os.path.join(directory, "chunks.npy"), mmap_mode="r")
    targets = np.load(os.path.join(directory, "references.npy"), mmap_mode="r")
    lengths = np.load(os.path.join(directory, "reference_lengths.npy"), mmap_mode="r")
    indices = os.path.join(directory, "indices.npy")
    if os.path.exists(indices):
        print("[indices.npy found: using idx for subsampling]")
        idx = np.load(indices, mmap_mode="r")
        idx = idx[idx < lengths.shape[0]]
        if limit:
            idx = idx[:limit]
        return chunks[idx, :], targets[idx, :], lengths[idx]
    if limit:
        chunks, targets, lengths = chunks[:limit], targets[:limit], lengths[:limit]
    return np.array(chunks), np.array(targets), np.array(lengths)


def load_validation(limit, directory):
    """The validation split load_numpy hands to the evaluator: validation/ when it exists, else the last 3 % of the
    (limited) training arrays."""
    directory = str(directory)
    sub = os.path.join(directory, "validation")
    if os.path.exists(sub):
        return load_numpy_datasets(directory=sub)
    train = load_numpy_datasets(limit=limit, directory=directory)
    print("[validation set not found: splitting training set (97%-3%)]")
    split = np.floor(len(train[0]) * 0.97).astype(np.int32)
    return tuple(x[split:] for x in train)
