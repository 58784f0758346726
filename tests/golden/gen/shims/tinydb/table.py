class Document(dict):
    pass
