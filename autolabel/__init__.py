"""Drop-in alias: ``import autolabel.models`` / ``autolabel.trainer`` / ``autolabel.dataset`` / ``autolabel.model_utils`` /
``autolabel.utils`` resolve to the MI355X-native implementations in ``autolabel_amd`` so that the reference's own
``scripts/train.py``, ``render.py`` and ``export.py`` import and run unchanged."""
import importlib
import sys

for _name in ['models', 'trainer', 'dataset', 'model_utils', 'utils', 'renderer', 'synthetic']:
    sys.modules[f'{__name__}.{_name}'] = importlib.import_module(f'autolabel_amd.{_name}')
    globals()[_name] = sys.modules[f'{__name__}.{_name}']
