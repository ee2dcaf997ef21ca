"""Drop-in for .../BERT/BertForSequenceClassification.py of the reference."""
import os as _os
import sys as _sys

_root = _os.path.abspath(_os.path.join(_os.path.dirname(__file__), "..", "..", "..", "..", ".."))
if _root not in _sys.path:
    _sys.path.insert(0, _root)
from transformer_explainability_amd import rules as _rules  # noqa: E402
from transformer_explainability_amd.bert import BertConfigLite, make_bert_module  # noqa: E402,F401

_ns = make_bert_module(_rules)
BertForSequenceClassification = _ns['BertForSequenceClassification']
BertModel = _ns['BertModel']
