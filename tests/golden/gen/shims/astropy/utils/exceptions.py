class ErfaWarning(Warning):
    pass
