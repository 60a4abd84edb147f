// na_json.cpp -- recursive-descent RFC 8259 reader (see na_json.h).
#include "na_json.h"

#include <algorithm>
#include <charconv>
#include <clocale>
#include <cstdlib>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace na
{
	class JsonParser
	{
	public:
		explicit JsonParser(const std::string& t) : p(t.data()), end(t.data() + t.size()) {}

		Json ParseDocument()
		{
			SkipWs();
			Json v = ParseValue(0);
			SkipWs();
			if (p != end) Fail("trailing characters");
			return v;
		}

	private:
		const char* p;
		const char* end;

		[[noreturn]] void Fail(const char* what) { throw std::runtime_error(std::string("json parse error: ") + what); }

		void SkipWs()
		{
			while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++;
		}

		Json ParseValue(int depth)
		{
			if (depth > 256) Fail("nesting too deep");
			if (p >= end) Fail("unexpected end");
			switch (*p)
			{
			case '{': return ParseObject(depth);
			case '[': return ParseArray(depth);
			case '"':
			{
				Json v;
				v.type = Json::String;
				v.str = ParseString();
				return v;
			}
			case 't':
				Literal("true");
				{
					Json v;
					v.type = Json::Bool;
					v.boolean = true;
					return v;
				}
			case 'f':
				Literal("false");
				{
					Json v;
					v.type = Json::Bool;
					v.boolean = false;
					return v;
				}
			case 'n':
				Literal("null");
				return Json();
			default:
				return ParseNumber();
			}
		}

		void Literal(const char* lit)
		{
			size_t n = strlen(lit);
			if ((size_t)(end - p) < n || memcmp(p, lit, n) != 0) Fail("bad literal");
			p += n;
		}

		Json ParseNumber()
		{
			const char* s = p;
			bool isInt = true;
			if (p < end && *p == '-') p++;
			if (p >= end || !((*p >= '0' && *p <= '9'))) Fail("bad number");
			while (p < end && *p >= '0' && *p <= '9') p++;
			if (p < end && *p == '.')
			{
				isInt = false;
				p++;
				if (p >= end || !(*p >= '0' && *p <= '9')) Fail("bad fraction");
				while (p < end && *p >= '0' && *p <= '9') p++;
			}
			if (p < end && (*p == 'e' || *p == 'E'))
			{
				isInt = false;
				p++;
				if (p < end && (*p == '+' || *p == '-')) p++;
				if (p >= end || !(*p >= '0' && *p <= '9')) Fail("bad exponent");
				while (p < end && *p >= '0' && *p <= '9') p++;
			}
			// std::from_chars is locale-independent (strtod follows LC_NUMERIC: under a comma-decimal locale "0.1234" would parse as 0)
			Json v;
			v.type = Json::Number;
			v.number = 0.0;
			const std::from_chars_result res = std::from_chars(s, p, v.number);
			if (res.ec == std::errc::invalid_argument || res.ptr != p) Fail("bad number");
			if (res.ec == std::errc::result_out_of_range)
			{
				// from_chars leaves the value untouched on overflow / underflow -- and libstdc++ 11 reports subnormal results as out of
				// range too.  Decide by the decimal magnitude of the literal (position of its first non-zero digit + exponent), which needs no
				// locale: far below DBL_MIN -> 0 (with the sign), far above DBL_MAX -> +-inf (what nlohmann keeps), in between (subnormals,
				// the edges) -> strtod on a copy whose decimal point is the current locale's.
				const bool neg = (*s == '-');
				const char* q = s + ((*s == '-') ? 1 : 0);
				long exp10 = 0, firstDigit = 0, intDigits = 0;
				bool seenNonZero = false, inFrac = false;
				long fracZeros = 0;
				for (; q < p && *q != 'e' && *q != 'E'; q++)
				{
					if (*q == '.') { inFrac = true; continue; }
					if (!seenNonZero)
					{
						if (*q != '0') { seenNonZero = true; firstDigit = inFrac ? -(fracZeros + 1) : 0; }
						else if (inFrac) fracZeros++;
					}
					if (!inFrac && seenNonZero) intDigits++;
				}
				if (q < p) exp10 = std::strtol(q + 1, nullptr, 10);
				// decimal exponent of the leading digit: d.ddd x 10^mag
				const long mag = exp10 + (intDigits > 0 ? intDigits - 1 : firstDigit);
				if (!seenNonZero || mag < -330) v.number = neg ? -0.0 : 0.0;
				else if (mag > 310) v.number = neg ? -HUGE_VAL : HUGE_VAL;
				else
				{
					std::string copy(s, p);
					const char point = *std::localeconv()->decimal_point;
					for (char& ch : copy)
						if (ch == '.') ch = point;
					v.number = std::strtod(copy.c_str(), nullptr); // subnormal or +-inf at the edges
				}
			}
			v.numberIsInteger = isInt;
			return v;
		}

		static void AppendUtf8(std::string& out, unsigned cp)
		{
			if (cp < 0x80) out.push_back((char)cp);
			else if (cp < 0x800)
			{
				out.push_back((char)(0xC0 | (cp >> 6)));
				out.push_back((char)(0x80 | (cp & 0x3F)));
			}
			else if (cp < 0x10000)
			{
				out.push_back((char)(0xE0 | (cp >> 12)));
				out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
				out.push_back((char)(0x80 | (cp & 0x3F)));
			}
			else
			{
				out.push_back((char)(0xF0 | (cp >> 18)));
				out.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
				out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
				out.push_back((char)(0x80 | (cp & 0x3F)));
			}
		}

		unsigned Hex4()
		{
			if (end - p < 4) Fail("bad \\u escape");
			unsigned v = 0;
			for (int i = 0; i < 4; i++)
			{
				char c = *p++;
				v <<= 4;
				if (c >= '0' && c <= '9') v |= (unsigned)(c - '0');
				else if (c >= 'a' && c <= 'f') v |= (unsigned)(c - 'a' + 10);
				else if (c >= 'A' && c <= 'F') v |= (unsigned)(c - 'A' + 10);
				else Fail("bad hex digit");
			}
			return v;
		}

		std::string ParseString()
		{
			std::string out;
			p++; // opening quote
			while (true)
			{
				if (p >= end) Fail("unterminated string");
				char c = *p++;
				if (c == '"') break;
				if (c == '\\')
				{
					if (p >= end) Fail("bad escape");
					char e = *p++;
					switch (e)
					{
					case '"': out.push_back('"'); break;
					case '\\': out.push_back('\\'); break;
					case '/': out.push_back('/'); break;
					case 'b': out.push_back('\b'); break;
					case 'f': out.push_back('\f'); break;
					case 'n': out.push_back('\n'); break;
					case 'r': out.push_back('\r'); break;
					case 't': out.push_back('\t'); break;
					case 'u':
					{
						unsigned cp = Hex4();
						if (cp >= 0xD800 && cp <= 0xDBFF && end - p >= 6 && p[0] == '\\' && p[1] == 'u')
						{
							p += 2;
							unsigned lo = Hex4();
							cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
						}
						AppendUtf8(out, cp);
						break;
					}
					default: Fail("unknown escape");
					}
				}
				else
				{
					out.push_back(c);
				}
			}
			return out;
		}

		Json ParseArray(int depth)
		{
			Json v;
			v.type = Json::Array;
			p++;
			SkipWs();
			if (p < end && *p == ']')
			{
				p++;
				return v;
			}
			while (true)
			{
				SkipWs();
				v.arr.push_back(ParseValue(depth + 1));
				SkipWs();
				if (p >= end) Fail("unterminated array");
				if (*p == ',')
				{
					p++;
					continue;
				}
				if (*p == ']')
				{
					p++;
					break;
				}
				Fail("expected , or ]");
			}
			return v;
		}

		Json ParseObject(int depth)
		{
			Json v;
			v.type = Json::Object;
			p++;
			SkipWs();
			if (p < end && *p == '}')
			{
				p++;
				return v;
			}
			while (true)
			{
				SkipWs();
				if (p >= end || *p != '"') Fail("expected key");
				std::string key = ParseString();
				SkipWs();
				if (p >= end || *p != ':') Fail("expected :");
				p++;
				SkipWs();
				Json val = ParseValue(depth + 1);
				if (v.obj.find(key) == v.obj.end()) v.keys.push_back(key);
				v.obj[key] = std::move(val);
				SkipWs();
				if (p >= end) Fail("unterminated object");
				if (*p == ',')
				{
					p++;
					continue;
				}
				if (*p == '}')
				{
					p++;
					break;
				}
				Fail("expected , or }");
			}
			return v;
		}
	};

	Json Json::Parse(const std::string& text)
	{
		JsonParser parser(text);
		return parser.ParseDocument();
	}

	static void DumpString(const std::string& s, std::string& out)
	{
		out.push_back('"');
		for (unsigned char c : s)
		{
			switch (c)
			{
			case '"': out += "\\\""; break;
			case '\\': out += "\\\\"; break;
			case '\b': out += "\\b"; break;
			case '\f': out += "\\f"; break;
			case '\n': out += "\\n"; break;
			case '\r': out += "\\r"; break;
			case '\t': out += "\\t"; break;
			default:
				if (c < 0x20)
				{
					char buf[8];
					snprintf(buf, sizeof(buf), "\\u%04x", c);
					out += buf;
				}
				else out.push_back((char)c);
			}
		}
		out.push_back('"');
	}

	static void DumpValue(const Json& v, std::string& out);

	std::string Json::Dump() const
	{
		std::string out;
		DumpValue(*this, out);
		return out;
	}

	static void DumpValue(const Json& v, std::string& out)
	{
		switch (v.GetType())
		{
		case Json::Null: out += "null"; break;
		case Json::Bool: out += v.AsBool() ? "true" : "false"; break;
		case Json::Number:
		{
			// nlohmann::json::dump(): integer tokens as integers, everything else in the shortest form that round-trips, with a
			// trailing ".0" when that form has neither a fraction nor an exponent (3.0 stays "3.0")
			const double d = v.AsDouble();
			char buf[48];
			if (!std::isfinite(d))
			{
				out += "null";
				break;
			}
			if (v.IsIntegerToken() && std::fabs(d) < 9.2e18)
			{
				snprintf(buf, sizeof(buf), "%lld", (long long)d);
				out += buf;
				break;
			}
			const std::to_chars_result r = std::to_chars(buf, buf + sizeof(buf) - 3, d);
			std::string tok(buf, r.ptr);
			if (tok.find_first_of(".eE") == std::string::npos) tok += ".0";
			out += tok;
			break;
		}
		case Json::String: DumpString(v.AsString(), out); break;
		case Json::Array:
			out.push_back('[');
			for (size_t i = 0; i < v.Size(); i++)
			{
				if (i) out.push_back(',');
				DumpValue(v.At(i), out);
			}
			out.push_back(']');
			break;
		case Json::Object:
		{
			out.push_back('{');
			bool first = true;
			std::vector<std::string> keys = v.Keys();
			std::sort(keys.begin(), keys.end()); // nlohmann::json objects are std::map: keys come out sorted
			for (const auto& k : keys)
			{
				if (!first) out.push_back(',');
				first = false;
				DumpString(k, out);
				out.push_back(':');
				DumpValue(v.At(k), out);
			}
			out.push_back('}');
			break;
		}
		}
	}

	void Json::FlattenNumbers(std::vector<float>& out) const
	{
		if (type == Array)
		{
			for (const auto& e : arr)
			{
				if (e.type == Array) e.FlattenNumbers(out);
				else out.push_back(e.AsFloat());
			}
		}
		else
		{
			out.push_back(AsFloat());
		}
	}
}
