from . import pointops
