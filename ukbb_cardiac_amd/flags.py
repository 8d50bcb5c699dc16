"""absl / ``tf.app.flags`` compatible command-line parsing for the drop-in
deployment scripts (reference: ``common/deploy_network.py:25-40``,
``common/deploy_network_ao.py:25-49``).

Accepted spellings, as absl does: ``--name value``, ``--name=value``, one or two
leading dashes; booleans ``--name``, ``--noname``, ``--name=true|false|1|0``.
Unknown flags raise; enum values are validated.
"""
from types import SimpleNamespace

_TRUE = ('true', 't', '1', 'yes', 'y')
_FALSE = ('false', 'f', '0', 'no', 'n')


class FlagError(ValueError):
    pass


class FlagSet:
    def __init__(self):
        self._defs = {}

    def DEFINE_string(self, name, default, help=''):
        self._defs[name] = ('string', default, None, help)

    def DEFINE_integer(self, name, default, help=''):
        self._defs[name] = ('integer', default, None, help)

    def DEFINE_float(self, name, default, help=''):
        self._defs[name] = ('float', default, None, help)

    def DEFINE_boolean(self, name, default, help=''):
        self._defs[name] = ('boolean', default, None, help)

    def DEFINE_enum(self, name, default, values, help=''):
        self._defs[name] = ('enum', default, list(values), help)

    def _convert(self, name, text):
        kind, _, values, _ = self._defs[name]
        try:
            if kind == 'integer':
                return int(text)
            if kind == 'float':
                return float(text)
        except ValueError:
            raise FlagError('flag --%s=%s: not a valid %s' % (name, text, kind))
        if kind == 'boolean':
            t = text.lower()
            if t in _TRUE:
                return True
            if t in _FALSE:
                return False
            raise FlagError('flag --%s=%s: not a boolean' % (name, text))
        if kind == 'enum' and text not in values:
            raise FlagError('flag --%s=%s: value should be one of <%s>' % (name, text, '|'.join(values)))
        return text

    def parse(self, argv):
        vals = {k: v[1] for k, v in self._defs.items()}
        rest = []
        i = 0
        args = list(argv)
        while i < len(args):
            a = args[i]
            i += 1
            if a == '--':
                rest.extend(args[i:])
                break
            if not a.startswith('-') or a == '-':
                rest.append(a)
                continue
            body = a.lstrip('-')
            name, eq, text = body.partition('=')
            if name in self._defs:
                kind = self._defs[name][0]
                if kind == 'boolean':
                    vals[name] = self._convert(name, text) if eq else True
                else:
                    if not eq:
                        if i >= len(args):
                            raise FlagError('flag --%s needs a value' % name)
                        text = args[i]
                        i += 1
                    vals[name] = self._convert(name, text)
            elif name.startswith('no') and name[2:] in self._defs and self._defs[name[2:]][0] == 'boolean' and not eq:
                vals[name[2:]] = False
            else:
                raise FlagError('unknown command line flag %r' % name)
        return SimpleNamespace(**vals), rest

    def usage(self):
        lines = []
        for name, (kind, default, values, help_) in self._defs.items():
            extra = ' <%s>' % '|'.join(values) if values else ''
            lines.append('  --%s%s (%s, default %r)\n      %s' % (name, extra, kind, default, help_))
        return '\n'.join(lines)
