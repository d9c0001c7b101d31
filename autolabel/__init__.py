"""Drop-in alias: ``import autolabel.models`` / ``autolabel.trainer`` / ``autolabel.dataset`` / ``autolabel.model_utils`` /
``autolabel.utils`` / ``autolabel.renderer`` resolve to the MI355X-native implementations in ``autolabel_amd``, so code written
against the reference's hot-path modules (its ``scripts/train.py`` and ``scripts/export.py``: they import exactly these) runs
on them unchanged.  NOT aliased -- out of the hot path: ``autolabel.constants``, ``autolabel.visualization``,
``autolabel.utils.feature_utils``, ``autolabel.backend``, ``autolabel.evaluation``; the reference's ``scripts/render.py`` imports
the first three for its colour maps / video writer and therefore needs the reference package for those (this repo's
``scripts/render.py`` restates the rendering part).  Depth frames of the eager loader are resized
bilinearly like the reference's (its ``cv2.resize(depth, size, cv2.INTER_NEAREST)`` passes the flag in the ``dst`` position:
autolabel/dataset.py:391-393; ``autolabel_amd.dataset._resize_linear``)."""
import importlib
import sys

for _name in ['models', 'trainer', 'dataset', 'model_utils', 'utils', 'renderer', 'synthetic']:
    sys.modules[f'{__name__}.{_name}'] = importlib.import_module(f'autolabel_amd.{_name}')
    globals()[_name] = sys.modules[f'{__name__}.{_name}']
