def load(path):  # import-only stand-in (NuRadioReco/__init__.py reads the version)
    return {'tool': {'poetry': {'name': 'NuRadioMC', 'version': '3.2.0-dev'}}}
