from . import function
