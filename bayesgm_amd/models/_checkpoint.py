"""Checkpoint directory management with the semantics of the reference's tf.train.CheckpointManager
(models/causalbgm/base.py:112-128, :524-530; models/bgm/base.py:122-139): checkpoints of one run live in
``{output_dir}/checkpoints/{dataset}/{timestamp}``, at most ``max_to_keep`` (5) of them are kept -- the oldest is deleted when a
sixth is saved -- and a model constructed on a directory that already holds checkpoints restores the latest one.  The files
are .npz archives (the reference writes TensorFlow checkpoint shards; the build stores the same information -- network
parameters, optimizer slots and step counters, the latent table with its slots -- as named arrays)."""
import json
import os

import numpy as np

INDEX = "checkpoint.json"


class CheckpointManager(object):
    def __init__(self, directory, max_to_keep=5):
        self.directory = directory
        self.max_to_keep = int(max_to_keep)

    def _index_path(self):
        return os.path.join(self.directory, INDEX)

    def _read_index(self):
        try:
            with open(self._index_path()) as f:
                names = json.load(f)["all"]
        except (OSError, ValueError, KeyError):
            return []
        return [n for n in names if os.path.exists(os.path.join(self.directory, n))]

    @property
    def latest_checkpoint(self):
        names = self._read_index()
        return os.path.join(self.directory, names[-1]) if names else None

    def save(self, name, arrays):
        """Write `arrays` to <directory>/<name> (atomically), list it as the latest, prune to max_to_keep."""
        os.makedirs(self.directory, exist_ok=True)
        path = os.path.join(self.directory, name)
        tmp = path + ".tmp.npz"
        np.savez(tmp, **arrays)
        os.replace(tmp, path)
        names = [n for n in self._read_index() if n != name] + [name]
        while len(names) > self.max_to_keep:
            old = names.pop(0)
            try:
                os.remove(os.path.join(self.directory, old))
            except OSError:
                pass
        with open(self._index_path() + ".tmp", "w") as f:
            json.dump({"all": names, "latest": name}, f)
        os.replace(self._index_path() + ".tmp", self._index_path())
        return path
