"""Drop-in package: ``importlib.import_module('model.' + name).InpaintGenerator()`` as in the
reference's test.py:117-118 / evaluate.py:45-46, served by the MI355X implementation."""
