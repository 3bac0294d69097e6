"""Model surface of the reference (/root/reference/models/__init__.py:16-24).

``build_model(args)`` dispatches on ``args.dataset_file`` exactly like the reference:
``vid_single`` -> single-frame Deformable-DETR (+ depth fusion), ``vid_multi`` -> TransVOD,
``vid_multi_plusplus`` -> TransVOD++.  Imports are deferred so that ``models.ops`` can be used
on its own.
"""


def build_model(args):
    if args.dataset_file == "vid_single":
        from .deformable_detr_single import build
    elif args.dataset_file == "vid_multi":
        from .deformable_detr_multi import build
    elif args.dataset_file == "vid_multi_plusplus":
        from .deformable_detr_multi_plusplus import build
    else:
        raise ValueError(f"unknown dataset_file {args.dataset_file!r}")
    return build(args)
