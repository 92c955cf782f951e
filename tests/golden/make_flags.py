#!/usr/bin/env python
"""Extract the reference CLI's flag table (name, default, action) as DATA -> tests/golden/flags.json.
Parses /root/reference/train_chaos.py with `ast` (the file cannot be imported here: cv2/skimage are absent)."""
import ast
import json
import os

REF = os.environ.get('PP_REFERENCE', '/root/reference')
tree = ast.parse(open(os.path.join(REF, 'train_chaos.py')).read())
flags = {}
for node in ast.walk(tree):
    if isinstance(node, ast.Call) and getattr(node.func, 'attr', '') == 'add_argument':
        name = node.args[0].value
        kw = {k.arg: k.value for k in node.keywords}
        d = {}
        if 'default' in kw:
            d['default'] = ast.literal_eval(kw['default'])
        if 'action' in kw:
            d['action'] = ast.literal_eval(kw['action'])
        if 'required' in kw:
            d['required'] = ast.literal_eval(kw['required'])
        if 'type' in kw:
            d['type'] = kw['type'].id
        flags[name] = d
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'flags.json')
json.dump(flags, open(out, 'w'), indent=1, sort_keys=True)
print(len(flags), 'flags ->', out)
