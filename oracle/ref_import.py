"""Import the REAL reference (read-only, /root/reference) in the build container.

TEST INFRASTRUCTURE, container-only: nothing that runs on the GPU box may call
this (the reference does not exist there).  Used by oracle/gen_golden.py to
produce tests/golden/ and by the optional container-only cross-check test.

The reference star-imports seven third-party modules that are absent here
(SURVEY.md §8(c)); they are registered as empty stubs.  ``torch_cluster`` is
stubbed with oracle/cluster.py's restated ``fps``/``knn`` (PARITY UNPINNED at
that boundary).
"""
import os
import sys
import types
import warnings

REFERENCE_ROOT = '/root/reference'


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, 'model'))


def load():
    """Returns a namespace with the reference modules: model, implicit, modules,
    point_transformer_layer, geometry, inference, loss, pipeline."""
    assert available(), 'reference not mounted'
    from . import cluster
    for name in ['open3d', 'cv2', 'imageio', 'seaborn', 'wandb', 'torchvision',
                 'torchvision.datasets', 'torchvision.models', 'torchvision.transforms',
                 'torchvision.utils']:
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    tc = types.ModuleType('torch_cluster')
    tc.fps = lambda src, batch=None, ratio=0.5, random_start=True: cluster.fps(
        src, batch, ratio, random_start)
    tc.knn = lambda x, y, k, batch_x=None, batch_y=None: cluster.knn(x, y, k, batch_x, batch_y)
    sys.modules['torch_cluster'] = tc
    cwd = os.getcwd()
    os.chdir(REFERENCE_ROOT)
    added = [REFERENCE_ROOT] + [os.path.join(REFERENCE_ROOT, d) for d in ('data', 'eval', 'model', 'utils')]
    sys.path[:0] = added
    # the reference's own top-level module names shadow nothing of ours while loading
    saved = {k: sys.modules.pop(k) for k in ['__init__', 'model', 'implicit', 'modules', 'geometry',
                                             'point_transformer_layer', 'inference', 'utils', 'args',
                                             'data', 'logvis', 'loss', 'pipeline'] if k in sys.modules}
    try:
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            import model as r_model
            import implicit as r_implicit
            import modules as r_modules
            import point_transformer_layer as r_ptl
            import geometry as r_geometry
            import inference as r_inference
            import loss as r_loss
            import pipeline as r_pipeline
    finally:
        os.chdir(cwd)
        for p in added:
            sys.path.remove(p)
    ns = types.SimpleNamespace(model=r_model, implicit=r_implicit, modules=r_modules,
                               point_transformer_layer=r_ptl, geometry=r_geometry,
                               inference=r_inference, loss=r_loss, pipeline=r_pipeline)
    # keep the reference modules reachable only through `ns`
    for k in ['model', 'implicit', 'modules', 'geometry', 'point_transformer_layer', 'inference',
              'utils', 'args', 'data', 'logvis', 'loss', 'pipeline', '__init__']:
        sys.modules.pop(k, None)
    sys.modules.update(saved)
    return ns
