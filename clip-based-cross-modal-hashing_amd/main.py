"""Method registry + CLI with the reference's surface (main.py:18-46): `trainers` maps the method
name to a Trainer class and `trainer(args, 0)` constructs AND runs it.  Methods are imported lazily
(the reference's eager imports make `python main.py` un-importable as shipped)."""
import argparse
import importlib

from argsbase import str2bool

_REGISTRY = {
    'DSPH': ("train.DSPH.hash_train", "DSPHTrainer"),
    'DCHMT': ("train.DCHMT.hash_train", "DCHMTTrainer"),
    'TwDH': ("train.TwDH.hash_train", "TwDHTrainer"),
    'DNPH': ("train.DNPH_TOMM.hash_train", "DNPHTOMMTrainer"),
    'MITH': ("train.MITH.hash_train", "MITHTrainer"),
    'DNpH': ("train.DNpH_TMM.hash_train", "DNpHTMMTrainer"),
    'DMsH_LN': ("train.DMsH_LN.hash_train", "DMsH_LNTrainer"),
    'DHaPH': ("train.DHaPH.hash_train", "DHaPHTrainer"),
}
_NOT_BUILT = ['DPBE', 'DDWSH', 'DDBH', 'DScPH', 'DPSIH', 'DGHDGH']


class _LazyTrainers(dict):
    def __missing__(self, key):
        if key in _REGISTRY:
            mod, cls = _REGISTRY[key]
            self[key] = getattr(importlib.import_module(mod), cls)
            return self[key]
        if key in _NOT_BUILT:
            raise NotImplementedError(f"method {key}: trainer not built in this round (see DESIGN.md scope table)")
        raise KeyError(key)

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default


trainers = _LazyTrainers()

if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--method", type=str, default='DSPH', help="Trainer method name")
    parser.add_argument("--dataset", type=str, default="flickr", help="name of dataset")
    parser.add_argument("--output-dim", type=int, default=16)
    parser.add_argument("--is-train", type=str2bool, default=True)
    args, _ = parser.parse_known_args()

    # under `python -m torch.distributed.run --nproc-per-node N main.py ...` this is one process per GPU: the trainer's `rank`
    # argument is the GPU id, as upstream; batches, evaluation sets and gradients are sharded / averaged in train/base.py
    import torch
    import dist_utils as du
    _, world, local = du.init_from_env()
    gpu = local % max(torch.cuda.device_count(), 1) if world > 1 else 0
    if world > 1 and torch.cuda.is_available():
        torch.cuda.set_device(gpu)
    trainer = trainers.get(args.method)
    trainer(args, gpu)
    if du.active():
        torch.distributed.destroy_process_group()
