"""Import shim (test infrastructure): the reference annotates tensors with jaxtyping, which is not installed here."""


class _Annot:
    def __class_getitem__(cls, item):
        return cls


class Float(_Annot):
    pass


class Int(_Annot):
    pass
