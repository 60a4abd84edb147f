// na_json.h -- small self-contained JSON DOM reader for .nam / keras model files.
//
// The reference parses model files with nlohmann::json (NeuralAudio/NeuralModel.cpp:330-336), a
// third-party header that is not part of the reference tree (empty submodule).  The model files
// are plain JSON objects with numbers, strings, arrays, null and booleans; this reader covers
// exactly RFC 8259 and keeps number tokens as doubles (weights are stored as float afterwards,
// like `std::vector<float>` conversion in the reference).
#pragma once

#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace na
{
	class Json
	{
	public:
		enum Type { Null, Bool, Number, String, Array, Object };

		Json() : type(Null) {}

		static Json Parse(const std::string& text);

		Type GetType() const { return type; }
		bool IsNull() const { return type == Null; }
		bool IsBool() const { return type == Bool; }
		bool IsNumber() const { return type == Number; }
		bool IsString() const { return type == String; }
		bool IsArray() const { return type == Array; }
		bool IsObject() const { return type == Object; }

		bool AsBool() const { Expect(Bool); return boolean; }
		double AsDouble() const { Expect(Number); return number; }
		float AsFloat() const { return (float)AsDouble(); }
		int AsInt() const { return (int)AsDouble(); }
		const std::string& AsString() const { Expect(String); return str; }

		size_t Size() const
		{
			if (type == Array) return arr.size();
			if (type == Object) return keys.size();
			return 0;
		}

		// array access
		const Json& At(size_t i) const
		{
			Expect(Array);
			if (i >= arr.size()) throw std::out_of_range("json: array index out of range");
			return arr[i];
		}

		const Json& Back() const
		{
			Expect(Array);
			if (arr.empty()) throw std::out_of_range("json: empty array");
			return arr.back();
		}

		// object access (like nlohmann's at(): throws when missing)
		bool Contains(const std::string& key) const { return type == Object && obj.find(key) != obj.end(); }

		const Json& At(const std::string& key) const
		{
			Expect(Object);
			auto it = obj.find(key);
			if (it == obj.end()) throw std::out_of_range("json: key '" + key + "' not found");
			return it->second;
		}

		// true when the number was written without fraction / exponent (nlohmann keeps such tokens as integers)
		bool IsIntegerToken() const { return type == Number && numberIsInteger; }

		// keys in document order
		const std::vector<std::string>& Keys() const { return keys; }

		// compact serialisation (used for GetMetadata values, NeuralModelImpl.h:85-94 uses json.dump())
		std::string Dump() const;

		// flatten nested numeric arrays depth-first (InternalModel.h:291-309 FlattenWeights)
		void FlattenNumbers(std::vector<float>& out) const;

	private:
		friend class JsonParser;

		void Expect(Type t) const
		{
			if (type != t) throw std::runtime_error("json: type mismatch");
		}

		Type type;
		bool boolean = false;
		double number = 0.0;
		bool numberIsInteger = false;
		std::string str;
		std::vector<Json> arr;
		std::map<std::string, Json> obj;
		std::vector<std::string> keys;
	};
}
